// kernels_gfa.hip -- the step columns of a GFA's P / W lines -> ItemTable, on the device.
//
// The reference turns every path line into item ids on the host: parse_path_seq_to_item_vec / parse_walk_seq_to_item_vec
// (src/graph_broker/util.rs:1021-1091) split the column at ',' (P: `name+`, `name-`) or at '>' / '<' (W) and look every
// segment name up in a HashMap<Vec<u8>, ItemId> (graph.rs:308-375 builds it from the S lines) -- one hash lookup per
// step, three to five passes over the text, which is where a real-file run spends its time (chr22: ~17 s).  The step
// columns are > 90 % of the bytes of a pangenome GFA and every step is independent of every other once it is known where
// the columns are, so: the raw bytes go to HBM as they are, the host only finds the lines (it needs the S and P headers
// anyway), and
//   k_tok_count  one wave per 16 KB of a column counts the steps that START in its piece (16 bytes per lane, separators found
//                by byte-parallel arithmetic),
//   (scan)       the counts become output offsets -- and, at the first piece of every path, the ItemTable's id_prefsum,
//   k_tok_emit   the same waves stage 1 KB of text at a time in LDS, list the starts, and convert the decimal names one start
//                per lane (no loop over digits) -- ids (and orientations) leave in step order, coalesced.
// Names must be decimal numbers: the id is the number itself (`nice: true`, graph.rs:224-229: the names are 1..N in file
// order) or comes out of a table indexed by the number that the host fills from the S lines.  Graphs with other names
// keep the host parser.  A step whose name is not a number, or a number the graph has no segment for, fails the call --
// the reference panics there (util.rs:1021).
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>

#include <rocprim/rocprim.hpp>

#include "name_table.hpp"
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
// first byte and after every ','; W column: after every '>' or '<'.  (An empty name -- ",," -- is still a step here, and
// fails in k_tok_emit like an unknown name.)
//
// Both kernels read the text 16 bytes per lane (one aligned 16-byte load: 1 KB per wave and load) and find the separators
// with byte-parallel arithmetic on the four dwords -- no lane looks at a single byte.

// 0x80 in every byte of x that equals the byte replicated in c4 (exact: no carries between bytes)
__device__ static inline uint32_t tok_eq(uint32_t x, uint32_t c4) {
    const uint32_t y = x ^ c4;
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
}
// separators of a step column: ',' (P) or '>' / '<' (W: 0x3E / 0x3C, one bit apart)
template <bool WALK>
__device__ static inline uint32_t tok_sep(uint32_t x) {
    return WALK ? tok_eq(x & ~0x02020202u, 0x3C3C3C3Cu) : tok_eq(x, 0x2C2C2C2Cu);
}
// 0xFF in the bytes [a, b) of a dword (a, b in 0..4, clamped)
__device__ static inline uint32_t tok_bytes(int a, int b) {
    a = a < 0 ? 0 : (a > 4 ? 4 : a);
    b = b < 0 ? 0 : (b > 4 ? 4 : b);
    const uint32_t hi = b >= 4 ? 0xFFFFFFFFu : ((1u << (8 * b)) - 1u), lo = a >= 4 ? 0xFFFFFFFFu : ((1u << (8 * a)) - 1u);
    return hi & ~lo;
}

typedef uint32_t tok_u32x4 __attribute__((ext_vector_type(4)));

// how many steps START in every piece: the separators in [max(b - 1, col_b), e - 1), and the column's first byte (P)
template <bool WALK>
__device__ static inline uint32_t tok_count_piece(const uint8_t *__restrict__ text, const TokPiece &t, uint32_t lane) {
    const uint64_t lo = t.b > t.col_b ? t.b - 1 : t.col_b, hi = t.e - 1;  // (t.e > t.b >= col_b: a piece is never empty)
    uint32_t n = 0;
    for (uint64_t pos = (lo & ~15ull) + lane * 16ull; pos < hi; pos += 1024) {
        const tok_u32x4 v = *reinterpret_cast<const tok_u32x4 *>(text + pos);
        uint32_t f[4] = {tok_sep<WALK>(v.x), tok_sep<WALK>(v.y), tok_sep<WALK>(v.z), tok_sep<WALK>(v.w)};
        if (pos < lo || pos + 16 > hi) {  // the first / last lanes of the range: only its bytes
            const int a = pos < lo ? (int)(lo - pos) : 0, b = pos + 16 > hi ? (int)(hi - pos) : 16;
#pragma unroll
            for (int w = 0; w < 4; ++w) f[w] &= tok_bytes(a - 4 * w, b - 4 * w);
        }
        n += (uint32_t)(__builtin_popcount(f[0]) + __builtin_popcount(f[1]) + __builtin_popcount(f[2]) + __builtin_popcount(f[3]));
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    return n + ((!WALK && t.b == t.col_b) ? 1u : 0u);
}

__global__ __launch_bounds__(256) void k_tok_count(const uint8_t *__restrict__ text, const uint64_t *__restrict__ piece_off,
                                                   const uint64_t *__restrict__ col_b, const uint64_t *__restrict__ col_e,
                                                   const uint8_t *__restrict__ is_walk, uint32_t n_paths, uint64_t n_pieces,
                                                   uint64_t *__restrict__ counts) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_pieces) return;
    const TokPiece t = tok_piece_of(c, piece_off, col_b, col_e, is_walk, n_paths);
    const uint32_t n = t.walk ? tok_count_piece<true>(text, t, lane) : tok_count_piece<false>(text, t, lane);
    if (lane == 0) counts[c] = n;
}

// The ids.  Per 1 KB of text (16 bytes per lane, the next KB already on its way): the bytes go to a wave-private LDS buffer,
// every lane marks the steps that START in its 16 bytes (separator flags moved up by one byte, the last flag of the lane
// below carried in), a prefix sum over the lanes numbers them, and every lane drops the positions of its starts into a
// compact list.  Then the wave takes the list 64 starts at a time, one per lane: the 12 bytes behind the start come out of
// LDS (four aligned dwords + a funnel shift), the run of digits is measured with byte-parallel flags, the same window is
// read again ENDING at the last digit so that the digits sit right-aligned whatever their number, and three 4-digit
// groups are converted with one multiplication each.  Ids leave in step order, one coalesced store per 64 steps.
// Strict like the reference's parser (src/graph_broker/util.rs:1021-1091 panics on all of these): a name must be 1..10
// decimal digits without a leading zero, a P step ends in '+' / '-' followed by ',' or the end of the column, a P
// column does not end in ',', a W column starts with '>' / '<'.
constexpr uint32_t TOK_LDS_PAD = 16;                    // bytes in front of the chunk (a right-aligned window may start before it)
constexpr uint32_t TOK_LDS_BUF = TOK_LDS_PAD + 1024 + 48;  // + the 32 bytes that follow the chunk + slack for the dword reads
constexpr uint32_t TOK_LDS_LIST = 512;                   // a W column can start a step every 2 bytes

// numeric names with something in front of the number (`s12`: pnx_gfa_steps.name_prefix): the same bytes for every segment
struct NamePrefix {
    uint32_t lo, hi;  // the bytes, little-endian, zero-padded
    uint32_t len;     // 0..8
};
// BYNAME: the names are looked up in the name table (name_table.hpp) by their bytes instead of being read as numbers
template <bool WALK, bool BYNAME>
__device__ static inline void tok_emit_piece(const uint8_t *__restrict__ text, const TokPiece &t, uint32_t lane, uint64_t out,
                                             const uint32_t *__restrict__ id_of_name, uint64_t n_names, const NameTab &names, const NamePrefix &pre,
                                             uint32_t n_nodes, uint32_t *__restrict__ items, uint8_t *__restrict__ backward, uint32_t &bad,
                                             uint8_t *buf, uint16_t *list) {
    const uint64_t base = t.b & ~15ull;
    auto load = [&](uint64_t pos) {  // 16 bytes at pos; the bytes from the end of the column on read as 0
        tok_u32x4 v = tok_u32x4{0, 0, 0, 0};
        if (pos < t.col_e) {
            v = *reinterpret_cast<const tok_u32x4 *>(text + pos);
            if (pos + 16 > t.col_e) {
                const int k = (int)(t.col_e - pos);
                v.x &= tok_bytes(0, k);
                v.y &= tok_bytes(0, k - 4);
                v.z &= tok_bytes(0, k - 8);
                v.w &= tok_bytes(0, k - 12);
            }
        }
        return v;
    };
    // separator flag (0x80) of the byte in front of the first chunk, if that byte belongs to the column
    uint32_t carry = 0;
    if (base > t.col_b) carry = tok_sep<WALK>((uint32_t)text[base - 1]) & 0x80u;
    carry = (uint32_t)__builtin_amdgcn_readfirstlane((int)carry);
    if (lane < 4) reinterpret_cast<uint32_t *>(buf)[lane] = 0;  // the pad in front: never a digit ...
    uint32_t last = base > 0 ? (uint32_t)text[base - 1] : 0u;   // ... but its last byte is the character in front of the chunk ('>' / '<' of a W step)
    last = (uint32_t)__builtin_amdgcn_readfirstlane((int)last);
    tok_u32x4 cur = load(base + lane * 16ull);
    for (uint64_t g0 = base; g0 < t.e; g0 += 1024) {
        const uint64_t pos = g0 + lane * 16ull;
        const tok_u32x4 nxt = load(pos + 1024);
        *reinterpret_cast<tok_u32x4 *>(buf + TOK_LDS_PAD + lane * 16u) = cur;
        if (lane < 2) *reinterpret_cast<tok_u32x4 *>(buf + TOK_LDS_PAD + 1024 + lane * 16u) = nxt;  // the 32 bytes that follow the chunk
        if (lane == 2) *reinterpret_cast<tok_u32x4 *>(buf + TOK_LDS_PAD + 1056) = tok_u32x4{0, 0, 0, 0};
        if (lane == 3) buf[TOK_LDS_PAD - 1] = (uint8_t)last;
        last = ((uint32_t)__builtin_amdgcn_readlane((int)cur.w, 63)) >> 24;
        // ---- starts in this lane's 16 bytes ----
        uint32_t sp[4] = {tok_sep<WALK>(cur.x), tok_sep<WALK>(cur.y), tok_sep<WALK>(cur.z), tok_sep<WALK>(cur.w)};
        if (pos < t.col_b) {  // bytes in front of the column are not part of it
            const int a = pos + 16 <= t.col_b ? 16 : (int)(t.col_b - pos);
#pragma unroll
            for (int w = 0; w < 4; ++w) sp[w] &= tok_bytes(a - 4 * w, 4);
        }
        uint32_t below = (uint32_t)__shfl_up((int)sp[3], 1);
        if (lane == 0) below = carry << 24;
        carry = ((uint32_t)__builtin_amdgcn_readlane((int)sp[3], 63)) >> 24;
        uint32_t st[4] = {(sp[0] << 8) | (below >> 24), (sp[1] << 8) | (sp[0] >> 24), (sp[2] << 8) | (sp[1] >> 24), (sp[3] << 8) | (sp[2] >> 24)};
        if (!WALK && pos <= t.col_b && t.col_b < pos + 16) st[(t.col_b - pos) >> 2] |= 0x80u << (8 * ((t.col_b - pos) & 3));  // a P column starts with a step
        if (pos < t.b || pos + 16 > t.e) {  // only the starts inside the piece are this wave's
            const int a = pos < t.b ? (pos + 16 <= t.b ? 16 : (int)(t.b - pos)) : 0, b = pos >= t.e ? 0 : (pos + 16 > t.e ? (int)(t.e - pos) : 16);
#pragma unroll
            for (int w = 0; w < 4; ++w) st[w] &= tok_bytes(a - 4 * w, b - 4 * w);
        }
        const uint32_t mine = (uint32_t)(__builtin_popcount(st[0]) + __builtin_popcount(st[1]) + __builtin_popcount(st[2]) + __builtin_popcount(st[3]));
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= (uint32_t)o) incl += up;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t slot = incl - mine;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t f = st[w];
            while (f) {
                const uint32_t k = (uint32_t)__builtin_ctz(f) >> 3;
                f &= f - 1u;
                list[slot++] = (uint16_t)(lane * 16u + 4u * (uint32_t)w + k);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- one start per lane ----
        for (uint32_t j0 = 0; j0 < total; j0 += 64) {
            const uint32_t j = j0 + lane;
            if (j < total) {
                const uint32_t p_name = TOK_LDS_PAD + list[j];  // byte of the name's first character in buf
                uint32_t p = p_name;
                bool pre_ok = true;
                if (!BYNAME && pre.len) {  // the prefix must stand there; the number starts behind it
                    const uint32_t *c = reinterpret_cast<const uint32_t *>(buf + (p & ~3u));
                    const uint32_t sc = (p & 3u) * 8u;
                    const uint32_t c0 = c[0], c1 = c[1], c2 = c[2];
                    const uint32_t w0 = __builtin_amdgcn_alignbit(c1, c0, sc), w1 = __builtin_amdgcn_alignbit(c2, c1, sc);
                    const uint32_t m0 = pre.len >= 4u ? 0xFFFFFFFFu : (1u << (8u * pre.len)) - 1u;
                    const uint32_t m1 = pre.len >= 8u ? 0xFFFFFFFFu : (pre.len > 4u ? (1u << (8u * (pre.len - 4u))) - 1u : 0u);
                    pre_ok = (((w0 ^ pre.lo) & m0) | ((w1 ^ pre.hi) & m1)) == 0u;
                    p += pre.len;
                }
                const uint32_t *d = reinterpret_cast<const uint32_t *>(buf + (p & ~3u));
                const uint32_t sh = (p & 3u) * 8u;
                if (BYNAME) {
                    // 20 bytes from the start: the name ends at the first separator (P: ',' W: '>' / '<') or at the end of the
                    // column (0); a P name is followed by its sign.  Up to 16 bytes of name are the key of the lookup.
                    const uint32_t g0 = d[0], g1 = d[1], g2 = d[2], g3 = d[3], g4 = d[4], g5 = d[5];
                    const uint32_t x[5] = {__builtin_amdgcn_alignbit(g1, g0, sh), __builtin_amdgcn_alignbit(g2, g1, sh), __builtin_amdgcn_alignbit(g3, g2, sh),
                                           __builtin_amdgcn_alignbit(g4, g3, sh), __builtin_amdgcn_alignbit(g5, g4, sh)};
                    uint32_t E = 20u;
#pragma unroll
                    for (int w = 4; w >= 0; --w) {
                        const uint32_t f = tok_sep<WALK>(x[w]) | tok_eq(x[w], 0u);
                        if (f) E = 4u * (uint32_t)w + ((uint32_t)__builtin_ctz(f) >> 3);
                    }
                    const uint32_t nlen = WALK ? E : E - 1u;  // (E == 0: an empty P name wraps to a length no name has)
                    uint32_t back, id = 0;
                    bool ok = E < 20u && nlen >= 1u && nlen <= 16u;
                    if (WALK) {
                        back = buf[p - 1] == '<';
                    } else {
                        const uint32_t sign = E ? (uint32_t)buf[p + E - 1u] : 0u;
                        ok = ok && (sign == '+' || sign == '-');
                        back = sign == '-';
                    }
                    if (ok) {
                        auto low = [](uint32_t k) { return k >= 4u ? 0xFFFFFFFFu : (k ? (1u << (8u * k)) - 1u : 0u); };
                        const uint32_t m0 = low(nlen), m1 = low(nlen > 4u ? nlen - 4u : 0u), m2 = low(nlen > 8u ? nlen - 8u : 0u),
                                       m3 = low(nlen > 12u ? nlen - 12u : 0u);
                        const unsigned long long k0 = ((unsigned long long)(x[1] & m1) << 32) | (x[0] & m0);
                        const unsigned long long k1 = ((unsigned long long)(x[3] & m3) << 32) | (x[2] & m2);
                        id = name_lookup(names, k0, k1);
                    }
                    // 1: malformed, 2: unknown segment, 4: a name of more than 16 bytes (no terminator within 20: the device route does not take it)
                    if (id == 0 || id > n_nodes) bad |= ok ? 2u : ((E >= 20u || nlen > 16u) && E != 0u ? 4u : 1u);
                    items[out + j] = id;
                    if (backward) backward[out + j] = (uint8_t)back;
                    continue;
                }
                const uint32_t d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
                const uint32_t x[3] = {__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh), __builtin_amdgcn_alignbit(d3, d2, sh)};
                // digits: bytes '0'..'9'; L = the number of leading ones among the 12 bytes
                uint32_t nd[3];
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const uint32_t y = x[w] ^ 0x30303030u;
                    nd[w] = (((y & 0x7F7F7F7Fu) + 0x76767676u) | y) & 0x80808080u;  // 0x80 where the byte is NOT a digit
                }
                const uint32_t L = nd[0] ? (uint32_t)__builtin_ctz(nd[0]) >> 3
                                         : (nd[1] ? 4u + ((uint32_t)__builtin_ctz(nd[1]) >> 3) : (nd[2] ? 8u + ((uint32_t)__builtin_ctz(nd[2]) >> 3) : 12u));
                // the 16 bytes that END four bytes behind the last digit: digits right-aligned in y[0..2], what follows in y[3]
                const uint32_t q = p + L - 12u;  // (>= 4: the pad; L = 0 reads the bytes in front of the name, masked away below)
                const uint32_t *e = reinterpret_cast<const uint32_t *>(buf + (q & ~3u));
                const uint32_t sq = (q & 3u) * 8u;
                const uint32_t e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3], e4 = e[4];
                const uint32_t y0 = __builtin_amdgcn_alignbit(e1, e0, sq), y1 = __builtin_amdgcn_alignbit(e2, e1, sq),
                               y2 = __builtin_amdgcn_alignbit(e3, e2, sq), y3 = __builtin_amdgcn_alignbit(e4, e3, sq);
                // keep the top L bytes of the 12: the last four digits in y2, the four before in y1, at most two more in y0
                auto top = [](uint32_t k) { return k >= 4u ? 0xFFFFFFFFu : (k ? 0xFFFFFFFFu << (32u - 8u * k) : 0u); };
                const uint32_t m2 = top(L), m1 = top(L > 4u ? L - 4u : 0u), m0 = top(L > 8u ? L - 8u : 0u);
                const uint32_t t2 = (y2 & m2) - (0x30303030u & m2), t1 = (y1 & m1) - (0x30303030u & m1), t0 = (y0 & m0) - (0x30303030u & m0);
                auto four = [](uint32_t tt) {  // bytes d0 d1 d2 d3 (d0 lowest) -> d0 * 1000 + d1 * 100 + d2 * 10 + d3
                    const uint32_t pr = ((tt * 2561u) >> 8) & 0x00FF00FFu;  // (d0 * 10 + d1) | (d2 * 10 + d3) << 16
                    return (pr & 0xFFFFu) * 100u + (pr >> 16);
                };
                const uint64_t v = (uint64_t)four(t0) * 100000000ull + (uint64_t)(four(t1) * 10000u + four(t2));
                const uint32_t first = x[0] & 0xFFu, term = y3 & 0xFFu, after = (y3 >> 8) & 0xFFu;
                bool ok = pre_ok && L >= 1u && L <= 10u && !(first == '0' && L > 1u) && v <= 0xFFFFFFFFull;
                uint32_t back;
                if (WALK) {
                    ok = ok && (term == '>' || term == '<' || term == 0);  // 0: the end of the column
                    back = buf[p_name - 1] == '<';
                } else {
                    ok = ok && (term == '+' || term == '-') && (after == ',' || after == 0);
                    back = term == '-';
                }
                uint32_t id = 0;
                if (ok) id = id_of_name ? (v < n_names ? id_of_name[v] : 0u) : (uint32_t)v;
                if (id == 0 || id > n_nodes) bad |= ok ? 2u : 1u;
                items[out + j] = id;
                if (backward) backward[out + j] = (uint8_t)back;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        out += total;
        cur = nxt;
    }
    // what no step shows: a P column that ends in a separator, a W column that does not begin with one
    if (lane == 0) {
        if (!WALK && t.e == t.col_e && text[t.col_e - 1] == ',') bad |= 1u;
        if (WALK && t.b == t.col_b && text[t.col_b] != '>' && text[t.col_b] != '<') bad |= 1u;
    }
}

template <bool BYNAME>
__global__ __launch_bounds__(256) void k_tok_emit(const uint8_t *__restrict__ text, const uint64_t *__restrict__ piece_off,
                                                  const uint64_t *__restrict__ col_b, const uint64_t *__restrict__ col_e,
                                                  const uint8_t *__restrict__ is_walk, uint32_t n_paths, uint64_t n_pieces,
                                                  const uint64_t *__restrict__ piece_out, const uint32_t *__restrict__ id_of_name,
                                                  uint64_t n_names, NameTab names, NamePrefix pre, uint32_t n_nodes, uint32_t *__restrict__ items,
                                                  uint8_t *__restrict__ backward, uint32_t *__restrict__ flags) {
    __shared__ __attribute__((aligned(16))) uint8_t buf_all[4][TOK_LDS_BUF];
    __shared__ uint16_t list_all[4][TOK_LDS_LIST];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + wave;
    if (c >= n_pieces) return;
    const TokPiece t = tok_piece_of(c, piece_off, col_b, col_e, is_walk, n_paths);
    uint32_t bad = 0;
    if (t.walk) tok_emit_piece<true, BYNAME>(text, t, lane, piece_out[c], id_of_name, n_names, names, pre, n_nodes, items, backward, bad, buf_all[wave], list_all[wave]);
    else tok_emit_piece<false, BYNAME>(text, t, lane, piece_out[c], id_of_name, n_names, names, pre, n_nodes, items, backward, bad, buf_all[wave], list_all[wave]);
    for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o);
    if (lane == 0 && bad) atomicOr(flags, bad);
}

// ---- the name table: one thread per S line puts its name in, a second pass makes every name find its own id ----
// name field of segment i: (name_off, name_len) as the caller gave them, or -- name_len == NULL: off holds the offsets of the S
// lines -- the bytes between "S\t" and the next tab (17: longer than a key can be)
__device__ static inline uint32_t name_field(const uint8_t *__restrict__ text, uint64_t text_bytes, const uint64_t *__restrict__ off,
                                             const uint8_t *__restrict__ len, uint32_t i, uint64_t &at) {
    if (len) {
        at = off[i];
        return len[i];
    }
    at = off[i] + 2;
    if (at > text_bytes || text[off[i] + 1] != '\t') return 0u;  // (no name field at all: malformed)
    uint32_t l = 0;
    while (l < 17u && at + l < text_bytes && text[at + l] != '\t' && text[at + l] != '\n' && text[at + l] != '\r') ++l;
    return l;
}
__global__ void k_names_insert(const uint8_t *__restrict__ text, uint64_t text_bytes, const uint64_t *__restrict__ name_off,
                               const uint8_t *__restrict__ name_len, uint32_t n_nodes, NameTab t, uint32_t *__restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    uint64_t at;
    const uint32_t len = name_field(text, text_bytes, name_off, name_len, i, at);
    if (len == 0u || len > 16u) {
        atomicOr(flags, len ? 4u : 1u);
        return;
    }
    unsigned long long k0, k1;
    name_key(text + at, len, k0, k1);
    uint64_t slot = name_hash(k0, k1) & t.mask;
    for (;;) {
        if (atomicCAS(&t.e[slot].id, 0u, i + 1u) == 0u) {
            t.e[slot].k0 = k0;
            t.e[slot].k1 = k1;
            return;
        }
        slot = (slot + 1) & t.mask;
    }
}
__global__ void k_names_verify(const uint8_t *__restrict__ text, uint64_t text_bytes, const uint64_t *__restrict__ name_off,
                               const uint8_t *__restrict__ name_len, uint32_t n_nodes, NameTab t, uint32_t *__restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    uint64_t at;
    const uint32_t len = name_field(text, text_bytes, name_off, name_len, i, at);
    if (len == 0u || len > 16u) return;
    unsigned long long k0, k1;
    name_key(text + at, len, k0, k1);
    if (name_lookup(t, k0, k1) != i + 1u) atomicOr(flags, 8u);  // the same name sits in an earlier slot: it occurs twice
}

// ---- L lines on the device (graph.rs:276-306): both ends of every line -> canonical edge; the distinct edges numbered by
// their FIRST line (duplicates skipped like the reference, graph.rs:296) ----
// name of a node: decimal (the id itself, or id_of_name[number]) or, with a name table, its bytes
struct NodeNames {
    const uint32_t *id_of_name;
    uint64_t n_names;
    NameTab tab;
    uint32_t by_name;
    NamePrefix pre;  // numeric names: what stands in front of the number
};
// the field [b, e) of the text (e: the tab behind it) as a node id; 0 = no such node; malformed -> bad |= 1
__device__ static inline uint32_t node_of_field(const uint8_t *__restrict__ text, uint64_t b, uint64_t e, const NodeNames &nn, uint32_t n_nodes, uint32_t &bad) {
    const uint64_t len = e - b;
    if (len == 0) {
        bad |= 1u;
        return 0;
    }
    if (nn.by_name) {
        if (len > 16) {
            bad |= 4u;
            return 0;
        }
        unsigned long long k0, k1;
        name_key(text + b, (uint32_t)len, k0, k1);
        return name_lookup(nn.tab, k0, k1);
    }
    if (nn.pre.len) {  // the prefix must stand in front of the number
        if (len <= nn.pre.len) {
            bad |= 1u;
            return 0;
        }
        for (uint32_t k = 0; k < nn.pre.len; ++k) {
            const uint32_t want = ((k < 4 ? nn.pre.lo : nn.pre.hi) >> (8 * (k & 3))) & 0xFFu;
            if (text[b + k] != want) {
                bad |= 1u;
                return 0;
            }
        }
        b += nn.pre.len;
    }
    if (e - b > 10 || (text[b] == '0' && e - b > 1)) {
        bad |= 1u;
        return 0;
    }
    uint64_t v = 0;
    for (uint64_t i = b; i < e; ++i) {
        const uint32_t c = text[i];
        if (c < '0' || c > '9') {
            bad |= 1u;
            return 0;
        }
        v = v * 10 + (c - '0');
    }
    if (nn.id_of_name) return v < nn.n_names ? nn.id_of_name[v] : 0u;
    return v <= n_nodes ? (uint32_t)v : 0u;
}
__global__ void k_links_parse(const uint8_t *__restrict__ text, uint64_t text_bytes, const uint64_t *__restrict__ link_off, uint64_t n_links,
                              NodeNames nn, uint32_t n_nodes, unsigned long long *__restrict__ l_key, uint32_t *__restrict__ flags) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_links) return;
    uint32_t bad = 0;
    auto field_end = [&](uint64_t b) {  // first tab (or line end) at or behind b
        uint64_t e = b;
        while (e < text_bytes && text[e] != '\t' && text[e] != '\n') ++e;
        return e;
    };
    const uint64_t a0 = link_off[k] + 2, a1 = field_end(a0);
    unsigned long long key = 0;
    // L <u> <o1> <v> <o2> ...: every field must end in a tab up to the second orientation
    if (link_off[k] + 2 > text_bytes || text[link_off[k]] != 'L' || a1 + 2 >= text_bytes || text[a1] != '\t' || text[a1 + 2] != '\t') bad |= 1u;
    else {
        const uint64_t b0 = a1 + 3, b1 = field_end(b0);
        if (b1 + 1 >= text_bytes || text[b1] != '\t') bad |= 1u;
        else {
            const uint32_t u = node_of_field(text, a0, a1, nn, n_nodes, bad), v = node_of_field(text, b0, b1, nn, n_nodes, bad);
            const uint8_t c1 = text[a1 + 1], c2 = text[b1 + 1];
            // an orientation is '+' or '-' and nothing else (the reference panics on anything else: Orientation::from_pm,
            // graph.rs:42-48): a malformed link must not become a backward edge and change the edge ids
            if ((c1 != '+' && c1 != '-') || (c2 != '+' && c2 != '-')) bad |= 1u;
            const uint32_t o1 = c1 == '+' ? 0u : 1u, o2 = c2 == '+' ? 0u : 1u;
            if (!bad && (u == 0 || u > n_nodes || v == 0 || v > n_nodes)) bad |= 2u;
            if (!bad) {  // Edge::canonical (graph.rs:142-148); the orientations ride in the two top bits (node ids < 2^30)
                unsigned long long uv;
                uint32_t oo;
                if (u > v || (u == v && o1 == 1u)) {
                    uv = ((unsigned long long)v << 32) | u;
                    oo = ((o2 ^ 1u) << 1) | (o1 ^ 1u);
                } else {
                    uv = ((unsigned long long)u << 32) | v;
                    oo = (o1 << 1) | o2;
                }
                key = uv | ((unsigned long long)oo << 62);
            }
        }
    }
    l_key[k] = key;
    if (bad) atomicOr(flags, bad);
}
// The L lines found on the device (pnx_gfa_steps.n_links == PNX_LINKS_FIND): a line starts at byte 0 or behind a '\n'; the
// ones that start with 'L' are links (graph.rs:276).  (The S lines of PNX_NAMES_FIND are found by the same kernel: letter4 =
// the letter in all four bytes.)  One workgroup per 4 KB of text, 16 bytes per lane, byte-parallel
// compares; the offsets come out in file order (counts per tile, a scan, then the same kernel writes).
constexpr uint32_t FIND_TILE = 4096;
template <bool EMIT>
__global__ __launch_bounds__(256) void k_links_find(const uint8_t *__restrict__ text, uint64_t lo16, uint64_t lo, uint64_t hi, uint32_t letter4,
                                                    uint32_t *__restrict__ tile_cnt, const uint32_t *__restrict__ tile_base,
                                                    uint64_t *__restrict__ out) {
    using Scan = rocprim::block_scan<uint32_t, 256>;
    __shared__ typename Scan::storage_type scan_mem;
    const uint64_t p = lo16 + (uint64_t)blockIdx.x * FIND_TILE + threadIdx.x * 16u;
    uint32_t m = 0;  // bit i: byte p + i starts an L line
    if (p < hi) {
        const uint4 x = *reinterpret_cast<const uint4 *>(text + p);
        const uint32_t w[4] = {x.x, x.y, x.z, x.w};
        uint32_t carry = (p == 0 || text[p - 1] == '\n') ? 0x80u : 0u;  // "the byte before is a line end", as a flag in bit 7
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t nl = tok_eq(w[k], 0x0A0A0A0Au);
            const uint32_t f = tok_eq(w[k], letter4) & ((nl << 8) | carry);
            carry = nl >> 24;
            m |= (((f >> 7) & 1u) | ((f >> 14) & 2u) | ((f >> 21) & 4u) | ((f >> 28) & 8u)) << (4 * k);
        }
        if (p < lo) m &= ~0u << (uint32_t)(lo - p);                 // (lo16 <= lo < lo16 + 16)
        if (p + 16 > hi) m &= (1u << (uint32_t)(hi - p)) - 1u;
    }
    const uint32_t c = __popc(m);
    uint32_t before = 0, total = 0;
    Scan().exclusive_scan(c, before, 0u, total, scan_mem);
    if (!EMIT) {
        if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
        return;
    }
    uint64_t at = (uint64_t)tile_base[blockIdx.x] + before;
    while (m) {
        const uint32_t i = __ffs(m) - 1;
        m &= m - 1;
        out[at++] = p + i;
    }
}

struct LinkTab {
    unsigned long long *key;  // 0 = empty
    uint32_t *first;          // smallest line with this edge
    uint64_t mask;
};
__device__ static inline uint64_t link_hash(unsigned long long key) {
    uint64_t x = key * 0x9FB21C651E98DF25ull;
    return x ^ (x >> 31);
}
__global__ void k_links_insert(const unsigned long long *__restrict__ l_key, uint64_t n_links, LinkTab t) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_links) return;
    const unsigned long long key = l_key[k];
    if (!key) return;
    uint64_t slot = link_hash(key) & t.mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(t.key + slot, 0ull, key);
        if (prev == 0ull || prev == key) {
            atomicMin(t.first + slot, (uint32_t)k);
            return;
        }
        slot = (slot + 1) & t.mask;
    }
}
__global__ void k_links_first(const unsigned long long *__restrict__ l_key, uint64_t n_links, LinkTab t, uint32_t *__restrict__ is_first) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_links) return;
    const unsigned long long key = l_key[k];
    uint32_t f = 0;
    if (key) {
        uint64_t slot = link_hash(key) & t.mask;
        while (t.key[slot] != key) slot = (slot + 1) & t.mask;
        f = t.first[slot] == (uint32_t)k ? 1u : 0u;
    }
    is_first[k] = f;
}
// the distinct edges by id (= 1 + the number of first lines before theirs): canonical ends and orientations, as pnx_gfa_steps.edge_uv / edge_oo
__global__ void k_links_emit(const unsigned long long *__restrict__ l_key, const uint32_t *__restrict__ is_first, const uint32_t *__restrict__ rank,
                             uint64_t n_links, unsigned long long *__restrict__ e_uv, uint8_t *__restrict__ e_oo) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_links || !is_first[k]) return;
    const uint32_t id = rank[k] + 1u;
    e_uv[id] = l_key[k] & 0x3FFFFFFFFFFFFFFFull;
    e_oo[id] = (uint8_t)(l_key[k] >> 62);
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

static NamePrefix name_prefix_of(const pnx_gfa_steps *g) {
    NamePrefix pre{0, 0, names_by_bytes(g) ? 0u : std::min<uint32_t>(g->name_prefix_len, 8u)};
    unsigned char b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t k = 0; k < pre.len; ++k) b[k] = (unsigned char)g->name_prefix[k];
    pre.lo = (uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24);
    pre.hi = (uint32_t)b[4] | ((uint32_t)b[5] << 8) | ((uint32_t)b[6] << 16) | ((uint32_t)b[7] << 24);
    return pre;
}

// the lines of the text that start with `letter`, inside the bytes [lo, hi) (hi == 0: the whole text) -> their offsets in
// file order (device array `off`, n of them)
static int find_lines(pnx_ctx *ctx, char letter, uint64_t lo_in, uint64_t hi_in, DevBuf &off, uint64_t &n, DevBuf &tile_cnt, DevBuf &tile_base,
                      DevBuf &tmp) {
    hipStream_t st = ctx->stream;
    n = 0;
    const uint64_t hi = hi_in ? std::min<uint64_t>(hi_in, ctx->gfa_text_bytes) : ctx->gfa_text_bytes;
    const uint64_t lo = std::min<uint64_t>(lo_in, hi), lo16 = lo & ~15ull;
    const uint64_t tiles = (hi - lo16 + FIND_TILE - 1) / FIND_TILE;
    if (tiles >= 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "%c lines on the device: the text range is too long", letter);
    int rc;
    if ((rc = ensure(ctx, tile_cnt, (tiles + 1) * 4)) || (rc = ensure(ctx, tile_base, (tiles + 1) * 4))) return rc;
    const uint8_t *text = (const uint8_t *)ctx->d_gfa_text.p;
    const uint32_t letter4 = 0x01010101u * (uint8_t)letter;
    uint32_t found = 0;
    if (tiles) {
        PNX_HIP(ctx, hipMemsetAsync((uint32_t *)tile_cnt.p + tiles, 0, 4, st));
        hipLaunchKernelGGL(k_links_find<false>, dim3((unsigned)tiles), dim3(256), 0, st, text, lo16, lo, hi, letter4, (uint32_t *)tile_cnt.p,
                           (const uint32_t *)nullptr, (uint64_t *)nullptr);
        size_t tmp_bytes = 0;
        PNX_HIP(ctx, rocprim::exclusive_scan(nullptr, tmp_bytes, (uint32_t *)tile_cnt.p, (uint32_t *)tile_base.p, 0u, (size_t)tiles + 1,
                                             rocprim::plus<uint32_t>(), st));
        if ((rc = ensure(ctx, tmp, tmp_bytes ? tmp_bytes : 8))) return rc;
        PNX_HIP(ctx, rocprim::exclusive_scan(tmp.p, tmp_bytes, (uint32_t *)tile_cnt.p, (uint32_t *)tile_base.p, 0u, (size_t)tiles + 1,
                                             rocprim::plus<uint32_t>(), st));
        PNX_HIP(ctx, hipMemcpyAsync(&found, (uint32_t *)tile_base.p + tiles, 4, hipMemcpyDeviceToHost, st));
        PNX_HIP(ctx, hipStreamSynchronize(st));
    }
    n = found;
    if ((rc = ensure(ctx, off, (n ? n : 1) * 8))) return rc;
    if (n) {
        hipLaunchKernelGGL(k_links_find<true>, dim3((unsigned)tiles), dim3(256), 0, st, text, lo16, lo, hi, letter4, (uint32_t *)nullptr,
                           (const uint32_t *)tile_base.p, (uint64_t *)off.p);
        PNX_HIP(ctx, hipGetLastError());
    }
    return PNX_OK;
}

// node2id in HBM (name_table.hpp) from the name fields of the S lines; kept in ctx->d_name_tab until the upload ends
static int gfa_name_table(pnx_ctx *ctx, const pnx_gfa_steps *g, NameTab &names) {
    const bool find = names_found_on_device(g);
    if (!find && !g->name_len) return ctx->fail(PNX_EINVAL, "pnx_gfa_steps: name_off without name_len");
    if (g->id_of_name) return ctx->fail(PNX_EINVAL, "pnx_gfa_steps: name_off and id_of_name are two ways to name the segments: pass one");
    const uint32_t n = g->n_nodes;
    hipStream_t st = ctx->stream;
    uint64_t slots = 1024;
    while (slots < 2ull * n) slots <<= 1;
    struct Scratch {
        DevBuf off, len, tile_cnt, tile_base, tmp;
        ~Scratch() {
            for (DevBuf *b : {&off, &len, &tile_cnt, &tile_base, &tmp}) release(*b);
        }
    } sc;
    int rc;
    if ((rc = ensure(ctx, ctx->d_name_tab, slots * sizeof(NameEntry))) || (rc = ensure(ctx, ctx->d_flags, 8 * sizeof(uint32_t)))) return rc;
    if (find) {  // the S lines, in file order: segment i + 1 is the i-th of them (graph.rs:323-351)
        uint64_t found = 0;
        if ((rc = find_lines(ctx, 'S', g->name_lo, g->name_hi, sc.off, found, sc.tile_cnt, sc.tile_base, sc.tmp))) return rc;
        if (found != n) return ctx->fail(PNX_EINVAL, "PNX_NAMES_FIND: %llu S lines in the text range, n_nodes = %u", (unsigned long long)found, n);
    } else {
        if ((rc = ensure(ctx, sc.off, ((size_t)n + 1) * 8)) || (rc = ensure(ctx, sc.len, (size_t)n + 1))) return rc;
        if (n) PNX_HIP(ctx, hipMemcpyAsync(sc.off.p, g->name_off, (size_t)n * 8, hipMemcpyHostToDevice, st));
        if (n) PNX_HIP(ctx, hipMemcpyAsync(sc.len.p, g->name_len, (size_t)n, hipMemcpyHostToDevice, st));
    }
    names.e = (NameEntry *)ctx->d_name_tab.p;
    names.mask = slots - 1;
    PNX_HIP(ctx, hipMemsetAsync(ctx->d_name_tab.p, 0, slots * sizeof(NameEntry), st));
    PNX_HIP(ctx, hipMemsetAsync(ctx->d_flags.p, 0, 8 * sizeof(uint32_t), st));
    uint32_t flags = 0;
    if (n) {
        const uint8_t *text = (const uint8_t *)ctx->d_gfa_text.p;
        const uint8_t *d_len = find ? (const uint8_t *)nullptr : (const uint8_t *)sc.len.p;
        hipLaunchKernelGGL(k_names_insert, dim3((n + 255) / 256), dim3(256), 0, st, text, ctx->gfa_text_bytes, (const uint64_t *)sc.off.p, d_len, n,
                           names, (uint32_t *)ctx->d_flags.p);
        hipLaunchKernelGGL(k_names_verify, dim3((n + 255) / 256), dim3(256), 0, st, text, ctx->gfa_text_bytes, (const uint64_t *)sc.off.p, d_len, n,
                           names, (uint32_t *)ctx->d_flags.p);
        PNX_HIP(ctx, hipGetLastError());
    }
    PNX_HIP(ctx, hipMemcpyAsync(&flags, ctx->d_flags.p, 4, hipMemcpyDeviceToHost, st));
    PNX_HIP(ctx, hipStreamSynchronize(st));
    if (flags & 1u) return ctx->fail(PNX_EINVAL, "a segment has an empty name");
    if (flags & 4u) return ctx->fail(PNX_ELIMIT, "a segment name is longer than 16 bytes: the device tokeniser does not take such names");
    if (flags & 8u) return ctx->fail(PNX_EINVAL, "a segment name occurs more than once in the GFA");
    return PNX_OK;
}

// The L lines of the text -> the distinct edges by id (device arrays in the layout of pnx_gfa_steps.edge_uv / edge_oo).
// Needs the text in HBM and, for names that are not numbers, the name table of this upload.
int gfa_links_to_edges(pnx_ctx *ctx, const pnx_gfa_steps *g, DevBuf &d_e_uv, DevBuf &d_e_oo, uint32_t &n_edges) {
    n_edges = 0;
    const bool find = !g->link_off && g->n_links == PNX_LINKS_FIND;
    uint64_t n = find ? 0 : g->n_links;
    if (n >= 0xFFFFFFFEull) return ctx->fail(PNX_ELIMIT, "more than 2^32-2 L lines");
    if (g->n_nodes >= (1u << 30)) return ctx->fail(PNX_ELIMIT, "L lines on the device: at most 2^30-1 segments");
    hipStream_t st = ctx->stream;
    struct Scratch {
        DevBuf off, key, tkey, tfirst, first, rank, tmp, names, tile_cnt, tile_base;
        ~Scratch() {
            for (DevBuf *b : {&off, &key, &tkey, &tfirst, &first, &rank, &tmp, &names, &tile_cnt, &tile_base}) release(*b);
        }
    } sc;
    if (find) {  // which lines are L lines: asked of the text itself, inside the byte range the caller names (0, 0: all of it)
        int rc0 = find_lines(ctx, 'L', g->link_lo, g->link_hi, sc.off, n, sc.tile_cnt, sc.tile_base, sc.tmp);
        if (rc0) return rc0;
    }
    uint64_t slots = 1024;
    while (slots < 2 * n) slots <<= 1;
    const size_t n1 = n ? n : 1;
    int rc;
    if ((rc = ensure(ctx, sc.off, n1 * 8)) || (rc = ensure(ctx, sc.key, n1 * 8)) || (rc = ensure(ctx, sc.tkey, slots * 8)) ||
        (rc = ensure(ctx, sc.tfirst, slots * 4)) || (rc = ensure(ctx, sc.first, n1 * 4)) || (rc = ensure(ctx, sc.rank, n1 * 4)) ||
        (rc = ensure(ctx, ctx->d_flags, 8 * sizeof(uint32_t))))
        return rc;
    NodeNames nn{nullptr, 0, NameTab{}, 0, name_prefix_of(g)};
    if (names_by_bytes(g)) {
        nn.by_name = 1;
        nn.tab.e = (NameEntry *)ctx->d_name_tab.p;
        uint64_t ns = 1024;
        while (ns < 2ull * g->n_nodes) ns <<= 1;
        nn.tab.mask = ns - 1;
        if (!nn.tab.e) return ctx->fail(PNX_EINVAL, "L lines by name: the name table of this upload is missing (internal error)");
    } else if (g->id_of_name) {
        if ((rc = ensure(ctx, sc.names, (g->n_names ? g->n_names : 1) * 4))) return rc;
        PNX_HIP(ctx, hipMemcpyAsync(sc.names.p, g->id_of_name, (size_t)g->n_names * 4, hipMemcpyHostToDevice, st));
        nn.id_of_name = (const uint32_t *)sc.names.p;
        nn.n_names = g->n_names;
    }
    if (n && !find) PNX_HIP(ctx, hipMemcpyAsync(sc.off.p, g->link_off, n * 8, hipMemcpyHostToDevice, st));
    PNX_HIP(ctx, hipMemsetAsync(ctx->d_flags.p, 0, 8 * sizeof(uint32_t), st));
    PNX_HIP(ctx, hipMemsetAsync(sc.tkey.p, 0, slots * 8, st));
    PNX_HIP(ctx, hipMemsetAsync(sc.tfirst.p, 0xFF, slots * 4, st));
    const LinkTab lt{(unsigned long long *)sc.tkey.p, (uint32_t *)sc.tfirst.p, slots - 1};
    const unsigned grid = (unsigned)((n + 255) / 256);
    uint32_t tail[2] = {0, 0};
    if (n) {
        const uint8_t *text = (const uint8_t *)ctx->d_gfa_text.p;
        hipLaunchKernelGGL(k_links_parse, dim3(grid), dim3(256), 0, st, text, ctx->gfa_text_bytes, (const uint64_t *)sc.off.p, n, nn, g->n_nodes,
                           (unsigned long long *)sc.key.p, (uint32_t *)ctx->d_flags.p);
        hipLaunchKernelGGL(k_links_insert, dim3(grid), dim3(256), 0, st, (const unsigned long long *)sc.key.p, n, lt);
        hipLaunchKernelGGL(k_links_first, dim3(grid), dim3(256), 0, st, (const unsigned long long *)sc.key.p, n, lt, (uint32_t *)sc.first.p);
        size_t tmp_bytes = 0;
        hipError_t e = rocprim::exclusive_scan(nullptr, tmp_bytes, (uint32_t *)sc.first.p, (uint32_t *)sc.rank.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), st);
        if (e == hipSuccess && (rc = ensure(ctx, sc.tmp, tmp_bytes ? tmp_bytes : 8)) == PNX_OK)
            e = rocprim::exclusive_scan(sc.tmp.p, tmp_bytes, (uint32_t *)sc.first.p, (uint32_t *)sc.rank.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), st);
        if (rc) return rc;
        if (e != hipSuccess) return ctx->fail(PNX_EHIP, "L lines: scan failed: %s", hipGetErrorString(e));
        PNX_HIP(ctx, hipMemcpyAsync(&tail[0], (uint32_t *)sc.rank.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
        PNX_HIP(ctx, hipMemcpyAsync(&tail[1], (uint32_t *)sc.first.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
    }
    uint32_t flags = 0;
    PNX_HIP(ctx, hipMemcpyAsync(&flags, ctx->d_flags.p, 4, hipMemcpyDeviceToHost, st));
    PNX_HIP(ctx, hipStreamSynchronize(st));
    if (flags & 1u) return ctx->fail(PNX_EINVAL, "malformed L line");
    if (flags & 4u) return ctx->fail(PNX_ELIMIT, "an L line names a segment of more than 16 bytes: the device route does not take such names");
    if (flags & 2u) return ctx->fail(PNX_EINVAL, "an L line names a segment the graph does not have");
    n_edges = tail[0] + tail[1];
    if ((rc = ensure(ctx, d_e_uv, ((size_t)n_edges + 1) * 8)) || (rc = ensure(ctx, d_e_oo, (size_t)n_edges + 1))) return rc;
    PNX_HIP(ctx, hipMemsetAsync(d_e_uv.p, 0, 8, st));
    PNX_HIP(ctx, hipMemsetAsync(d_e_oo.p, 0, 1, st));
    if (n) hipLaunchKernelGGL(k_links_emit, dim3(grid), dim3(256), 0, st, (const unsigned long long *)sc.key.p, (const uint32_t *)sc.first.p,
                              (const uint32_t *)sc.rank.p, n, (unsigned long long *)d_e_uv.p, (uint8_t *)d_e_oo.p);
    PNX_HIP(ctx, hipGetLastError());
    PNX_HIP(ctx, hipStreamSynchronize(st));  // (the scratch goes with this scope)
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
    NameTab names;
    if (names_by_bytes(g)) {
        if ((rc = gfa_name_table(ctx, g, names))) return rc;
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
    if (n_pieces) {
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, text, (const uint64_t *)sc.off.p, (const uint64_t *)sc.cb.p,
                               (const uint64_t *)sc.ce.p, (const uint8_t *)sc.walk.p, P, n_pieces, (const uint64_t *)sc.outs.p, d_names, g->n_names,
                               names, name_prefix_of(g), g->n_nodes, (uint32_t *)ctx->d_items.p, d_backward ? (uint8_t *)d_backward->p : (uint8_t *)nullptr,
                               (uint32_t *)ctx->d_flags.p);
        };
        if (names_by_bytes(g)) go(k_tok_emit<true>);
        else go(k_tok_emit<false>);
    }
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    ctx->h_path_off.assign(p1, 0);
    uint32_t flags = 0;
    PNX_HIP(ctx, hipMemcpyAsync(ctx->h_path_off.data(), ctx->d_path_off.p, p1 * 8, hipMemcpyDeviceToHost, st));
    PNX_HIP(ctx, hipMemcpyAsync(&flags, ctx->d_flags.p, 4, hipMemcpyDeviceToHost, st));
    PNX_HIP(ctx, hipStreamSynchronize(st));
    if (flags & 1u) return ctx->fail(PNX_EINVAL, "a path step is not of the form <name><+|-> (P) / <'>'|'<'><name> (W)%s", names_by_bytes(g) ? "" : " with a decimal name");
    if (flags & 4u) return ctx->fail(PNX_ELIMIT, "a path step names a segment of more than 16 bytes: the device tokeniser does not take such names");
    if (flags & 2u) return ctx->fail(PNX_EINVAL, "a path step names a segment the graph does not have");
    ctx->n_steps = total;
    return PNX_OK;
}

}  // namespace pnx

namespace pnx {
// pnx_preload: the first launch of a kernel loads the code object of its translation unit (tens of ms) and builds the
// kernel's function object; asking for a kernel's attributes does the same, without a launch -- and can be done by a host
// thread that has nothing else to do while the GFA text travels to HBM
void preload_gfa(unsigned what) {
    hipFuncAttributes a;
    auto touch = [&a](const void *k) { (void)hipFuncGetAttributes(&a, k); };
    if (what & PNX_PRELOAD_GFA) {
        touch((const void *)k_tok_count);
        touch((const void *)k_tok_emit<false>);
        touch((const void *)k_tok_emit<true>);
        touch((const void *)k_tok_total);
        touch((const void *)k_tok_path_off);
        touch((const void *)k_names_insert);
        touch((const void *)k_names_verify);
    }
    if (what & PNX_PRELOAD_LINKS) {
        touch((const void *)k_links_find<false>);
        touch((const void *)k_links_find<true>);
        touch((const void *)k_links_parse);
        touch((const void *)k_links_insert);
        touch((const void *)k_links_first);
        touch((const void *)k_links_emit);
    }
}
}  // namespace pnx
