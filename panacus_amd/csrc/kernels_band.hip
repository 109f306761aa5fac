// kernels_band.hip -- the one-shot coverage pass: steps -> coverage vector + histogram in ONE read of the ItemTable (gfx950).
//
// AbacusByTotal::coverage (src/graph_broker/abacus.rs:719-744) sweeps the ItemTable once per abacus, and nothing in the
// reference ever sweeps the same table twice with the same grouping (graph_broker.rs:389-432): the unit of work is ONE
// sweep.  The path rows of kernels_rows.hip pay for a derived table (steps read once, rows written, rows read) that only
// a second sweep gets anything out of.  This file is the route for the first sweep of a graph whose paths run through the
// ids in order (pangenome paths of graphs with sorted ids; ascending or descending):
//
//   k_band_index   per (path of the visiting order, band edge): the step position where the path crosses the edge of a
//                  BAND of BT item tiles -- an interpolation search with sqrt-steps, a handful of dependent probes.
//   k_band_cover   one WORKGROUP owns one band (BT tiles = BT waves, wave w keeps the bit-sliced counters of tile w).
//                  The visiting order is taken CW entries at a time: wave w streams the whole band segment of entry
//                  k0 + w -- contiguous steps, 16-byte non-temporal loads, BAND_D of them in flight per lane, the next
//                  segment's first loads issued before the current one is folded -- and ORs presence bits into the
//                  entry's band bitmap in LDS (ds_or_b32: visiting an item twice is idempotent, which is the
//                  reference's last[] array); after ONE barrier per CW entries every wave reads its tile's slice of the
//                  CW bitmaps in visiting order and folds finished groups into its counters (tile_counters.hpp).  The
//                  tail of a tile -- coverage vector, histogram bins -- is the one the rows kernel has.
//
//   k_band_tail    what the pass ends with: the steps that were NOT in the band they were dealt to (below), then the sum of
//                  the histogram replicas, handed to the host.
//
// Exactness does not rest on the search, nor on the paths being sorted.  AbacusByTotal::coverage is insensitive to the order
// of the steps inside a group (abacus.rs:727-742: last[sid] != group): all that has to hold is that every step of every
// path is seen.  The segments of a path are the intervals between consecutive edge positions, whatever the searches
// returned: a chain of positions from 0 to the path's length covers every step at least once (twice does no harm: a
// presence bit is idempotent).  Every step is checked against the band it was dealt to; a step outside -- a local inversion
// across a band edge, a back-jump, a duplication elsewhere in the graph -- is SPILLED: (group, id) goes to a list in HBM
// (one wave-aggregated atomic per 256 steps that hold any), and the counters of the pass are those of the in-band steps.
// k_band_tail takes the list 64 records at a time: the records of one (group, band) cell -- consecutive steps of a path
// that left its band usually land in ONE cell -- go into a bitmap of that band; the wave streams the segments the group's
// paths have in that band and clears what it meets (those visits were counted in-band); what is left is deduplicated
// across the whole list by a hash set in HBM and added: coverage vector +1, histogram bin moved, presence bit set.  Spills
// cost what they read: a cell is a few thousand steps.  The list is bounded and so is the volume of the scans; beyond
// either bound (a graph whose paths do not follow the ids at all) flags[5] is raised: the pass is void and the host runs
// it again over path rows, which serve any path (pnx_api.hip: settle_oldest).  Steps ORed from positions of the SAME
// path beyond a segment's ends (16-byte alignment) are never foreign facts: positions outside the segment are masked.
//
// Small graphs (a node-range shard of a multi-GPU run): a band per workgroup would leave most of the chip idle, so the
// visiting order is cut at group boundaries into `splits`, one workgroup per (band, split); the counters of a band's
// splits meet in the coverage vector (atomic adds onto zeros) and the histogram is taken from it (kernels_hist.hip).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "pnx_context.hpp"
#include "step_chunks.hpp"
#include "tile_counters.hpp"

namespace pnx {

constexpr int BAND_CW = 4;                  // waves per workgroup = item tiles per band
constexpr uint32_t BAND_SHIFT = 13;         // log2 of the ids of a band (BAND_CW tiles of 2048)
static_assert((uint32_t)BAND_CW * BLOCK_ITEMS == 1u << BAND_SHIFT, "a band is 2^BAND_SHIFT ids");
constexpr uint64_t BAND_DESC = 1ull << 63;  // index entry: the path runs through the ids downwards
constexpr int BAND_MAX_SPLITS = 16;         // workgroups that share a band (each takes a range of the visiting order)
constexpr int SCAN_U = 4;                   // 16-byte loads a lane of k_band_tail keeps in flight while it scans a segment
constexpr uint32_t BAND_TAIL_WAVES = 8;     // waves per workgroup of k_band_tail (a wave takes one burst of the spill list at a time)
constexpr uint32_t BAND_TAIL_GRID = 1024;   // ... and at most this many workgroups (as many as the chip holds at once: launch_band_tail)

constexpr uint32_t LOOSE_MAX = 16;          // groups with paths that do not follow the ids at all which one pass takes in
constexpr uint32_t LOOSE_WORKGROUPS = 256;  // the first workgroups of k_band_tail's grid: they mark the steps of such groups and fold the bitmaps

// Groups with a path that does not follow the ids AT ALL (a shuffled path; edge ids in the order of the L lines): no position
// deals its steps to bands.  The index kernel recognises such a path by the sectors it samples along it (most of them hold
// steps that go up AND down), skips the searches and flags the path's GROUP and every entry of it; the coverage kernel leaves
// the flagged entries out; the first LOOSE_WORKGROUPS workgroups of the tail kernel stream those groups' steps, setting presence
// bits in a bitmap per group (the block layout of a row of the presence matrix; 1.25 MB for 10 M ids: the atomics stay in the
// L2s), wait for each other, and fold the bitmaps into the coverage vector and the histogram, tile by tile, clearing them.
// AbacusByTotal::coverage (abacus.rs:727-742) counts a group once per item however its steps are ordered: the result is exact.
// More than LOOSE_MAX such groups, or more steps in them than an eighth of the graph (one atomic per step is no way to read a
// graph): the pass is void and the path rows take over.
// state (= the probe statistics' block): [0] probes, [1] probes astray, [2] flagged groups, [3] their steps / 1024, [4 .. 4 + LOOSE_MAX) which ones;
// the tail kernel of every pass sets it, and the flags of the groups, back to zero.
struct BandLoose {
    uint32_t *state;
    uint32_t *group_flag;  // n_groups words: nonzero = every entry of the group is left to the bitmaps
    uint32_t *entry_flag;  // n_ordered words: the same per entry of the visiting order (what the coverage kernel reads)
    uint32_t *bits;        // LOOSE_MAX bitmaps of n_tiles x 64 words, zero between passes
};

// the visiting order cut at group boundaries: split s takes the entries [k[s], k[s + 1])
struct BandSplits {
    uint32_t n;
    uint32_t k[BAND_MAX_SPLITS + 1];
};

// where the steps go that were not in the band they were dealt to.  A BURST is what one wave found in one load (up to 256
// consecutive steps of one path, i.e. of one group): its ids go to rec[start ..], and dir[b] = (start << 32) | group says where
// they begin -- bursts are numbered in the order of their starts, so burst b ends where burst b + 1 begins.  flags[6] counts
// the records, flags[7] the bursts (one 64-bit atomic takes a range of both).
struct BandSpill {
    uint32_t *rec;
    unsigned long long *dir;
    uint32_t cap, dir_cap;
};

// first j in [0, len] with key(j) >= X, key = id (ascending path) or ~id (descending); keys are non-decreasing on a
// sorted path -- on any other the result is some position in [0, len]: the steps it deals to the wrong band are spilled.
// A probe is one aligned 64-byte sector = 16 steps (four 16-byte loads, one latency): either the sector holds the
// crossing, or its nearest step becomes one of the two points of the next secant guess (the path's ends at first: ids
// along a pangenome path are close to evenly spread, so the first guess is off by a few thousand steps of millions, the
// second by tens, the third lands in the sector) and tightens the bracket; after 8 probes the guess is the midpoint.
// The chain of dependent reads is what this kernel costs: ~4 probes here against ~14 single-step probes of a plain
// interpolation + binary search (0.058 -> 0.03 ms on 10 M items x 256 paths).
template <bool DESC>
__device__ static inline uint64_t band_edge_search(const uint32_t *__restrict__ items, uint64_t ps, uint64_t len, uint32_t X, uint32_t ka,
                                                   uint32_t kz, uint32_t &n_probes, uint32_t &n_astray) {
    if (ka >= X) return 0;
    if (kz < X) return len;
    uint64_t lo = 0, hi = len - 1;  // key(lo) < X <= key(hi)
    uint32_t klo = ka, khi = kz;    // ... those two keys: what a step between the two positions of a sorted path lies between
    double pa = 0.0, va = (double)ka, pb = (double)(len - 1), vb = (double)kz;  // the two points of the secant
    uint32_t astray_here = 0;  // (a search that keeps meeting steps from elsewhere is not converging on anything: any position will do)
    for (int iter = 0; hi - lo > 1 && astray_here < 8u; ++iter) {
        uint64_t g = lo + ((hi - lo) >> 1);
        if (iter < 8 && vb != va) {
            const double t = pb + ((double)X - vb) * (pb - pa) / (vb - va);
            g = t <= (double)(lo + 1) ? lo + 1 : (t >= (double)(hi - 1) ? hi - 1 : (uint64_t)t);
        }
        const uint64_t a0 = (ps + g) & ~15ull;  // the sector of step g (absolute index of its first step)
        const uint4 *src = reinterpret_cast<const uint4 *>(items + a0);
        const uint4 w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
        const uint32_t w[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
        // steps of the sector that belong to the path and lie inside the bracket: [i0, i1]
        const uint64_t first = ps + lo + 1, last = ps + hi - 1;  // absolute; first <= ps + g <= last
        const uint32_t i0 = first > a0 ? (uint32_t)(first - a0) : 0u, i1 = last - a0 < 15 ? (uint32_t)(last - a0) : 15u;
        uint32_t below = 0, k_first = 0, k_last = 0;
        bool astray = false;  // a step of the sector that cannot stand between the bracket's ends on a sorted path
#pragma unroll
        for (uint32_t i = 0; i < 16; ++i) {
            const uint32_t key = DESC ? ~w[i] : w[i];
            if (i == i0) k_first = key;
            if (i == i1) k_last = key;
            below += (i >= i0 && i <= i1 && key < X) ? 1u : 0u;
            astray |= i >= i0 && i <= i1 && (key < klo || key > khi);
        }
        n_probes += 1;
        n_astray += astray ? 1u : 0u;
        astray_here += astray ? 1u : 0u;
        const uint32_t n = i1 - i0 + 1;
        if (below != 0 && below != n) return a0 + i0 + below - ps;  // the crossing lies in the sector
        pa = pb;
        va = vb;
        if (below == 0) {  // all at or above X
            hi = a0 + i0 - ps;
            pb = (double)hi;
            vb = (double)k_first;
            khi = k_first;
        } else {
            lo = a0 + i1 - ps;
            pb = (double)lo;
            vb = (double)k_last;
            klo = k_last;
        }
    }
    return hi;
}

// bidx[k * (n_bands + 1) + e] = absolute step position where entry k's path crosses band edge e (id e * band_items), e = 0 ..
// n_bands, | BAND_DESC for a descending path: band b of an ascending path is [bidx[k][b], bidx[k][b+1]), of a descending one
// [bidx[k][b+1], bidx[k][b]).  The lanes of a wave take consecutive edges of ONE path: their probes stay within a few
// hundred KB of each other (one or two translation entries per round of probes instead of 64).  (Also clears the pass's
// counter block -- a memset in front is one more kernel in the chain --, lays out where the entries of every group begin
// (group_first, n_groups + 1: the tail kernel walks the paths of a group), and, for a pass whose bands are shared by several
// workgroups, zeroes the coverage vector they add to.)
// Which way a path runs: by the majority of five evenly spaced steps, its two ends deciding a tie -- a first or last step
// out of place does not turn the whole path around.
__global__ __launch_bounds__(256) void k_band_index(const uint32_t *__restrict__ items, const uint64_t *__restrict__ ent_start,
                                                    const uint64_t *__restrict__ ent_len, const uint32_t *__restrict__ ord_group,
                                                    uint32_t n_ordered, uint32_t n_groups, uint32_t n_bands, uint32_t band_items,
                                                    unsigned long long *__restrict__ bidx, uint32_t *__restrict__ group_first,
                                                    uint4 *__restrict__ block16, uint32_t n_block16, uint4 *__restrict__ zero16,
                                                    uint64_t n_zero16, uint32_t *__restrict__ probe_stats, uint32_t *__restrict__ group_loose,
                                                    uint32_t *__restrict__ entry_loose, unsigned long long *__restrict__ dbg_time) {
    // PNX_BAND_INDEX_TIMING (measurement): one wave in the middle of the grid stamps the 100 MHz clock behind every phase
    const bool dbg_wave = dbg_time && blockIdx.x == gridDim.x / 2 && threadIdx.x < 64;
    auto stamp = [&](int i) {
        if (dbg_wave && threadIdx.x == 0) dbg_time[i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_threads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = tid; q < n_block16; q += n_threads) block16[q] = make_uint4(0, 0, 0, 0);
    for (uint64_t q = tid; q < n_zero16; q += n_threads) zero16[q] = make_uint4(0, 0, 0, 0);
    if (tid < n_ordered) {
        const uint32_t g = ord_group[tid];
        if (tid == 0 || ord_group[tid - 1] != g) group_first[g] = (uint32_t)tid;
        if (tid == 0) group_first[n_groups] = n_ordered;
    }
    const uint64_t total = (uint64_t)(n_bands + 1) * n_ordered;
    // (no early exit: the lanes of a wave compare notes at the end)
    const bool live = tid < total;
    const uint32_t e = live ? (uint32_t)(tid % (n_bands + 1)) : 0u, k = live ? (uint32_t)(tid / (n_bands + 1)) : 0xFFFFFFFFu;
    uint64_t ps = 0, len = 0;
    bool desc = false;
    uint64_t j = 0;
    uint32_t n_probes = 0, n_astray = 0;
    if (live) {
        ps = ent_start[k];
        len = ent_len[k];
    }
    if (dbg_wave && (ps + len) == 0x7FFFFFFFFFFFFFFFull) dbg_time[15] = 1;  // (a use of the loads: the stamp waits for them)
    stamp(1);
    // ---- does the path follow the ids at all?  Every lane looks at one 64-byte sector of its path (the lanes of a wave hold
    // consecutive edges of one path: their sectors are spread evenly along it): on a path that runs through the ids -- upwards
    // or downwards, with blocks reversed, repeated or from elsewhere -- the 15 steps from one id to the next inside a sector go
    // ONE way; where more than a quarter of them go the other way the sector is jumbled, and a path most of whose sectors are
    // is LOOSE: no search (each would take its full 30 probes, and mean nothing), its group flagged (BandLoose).  The verdict
    // is per wave; a path whose waves disagree is loose, since ANY wave's flag takes the whole group out of the bands.
    bool sampled = false, jumbled = false;
    // (the path's two ends and three inner samples -- what the direction and the bracket of the search are made of below -- are
    // asked for HERE, beside the sector: five more loads in the same round trip instead of a round trip of their own behind the
    // wave's verdict on the sectors; every access at a fresh address is ~5 us on this part)
    uint32_t end_a = 0, end_z = 0, in_q1 = 0, in_q2 = 0, in_q3 = 0;
    if (len) {
        end_a = items[ps];
        end_z = items[ps + len - 1];
        if (len >= 16) {
            in_q1 = items[ps + len / 4];
            in_q2 = items[ps + len / 2];
            in_q3 = items[ps + len / 2 + len / 4];
        }
    }
    if (len >= 1024) {
        const uint64_t pos = (uint64_t)e * (len - 16) / n_bands;
        uint64_t a0 = (ps + pos) & ~15ull;
        if (a0 < ps) a0 += 16;
        const uint4 *src = reinterpret_cast<const uint4 *>(items + a0);
        const uint4 w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
        const uint32_t w[16] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w};
        uint32_t up = 0, down = 0;
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            up += w[i + 1] > w[i] ? 1u : 0u;
            down += w[i + 1] < w[i] ? 1u : 0u;
        }
        sampled = true;
        jumbled = up + down >= 8u && (up < down ? up : down) * 4u > up + down;
    }
    if (dbg_wave && (end_a ^ end_z ^ in_q1 ^ in_q2 ^ in_q3) == 0xFFFFFFFEu && jumbled) dbg_time[15] = 2;
    stamp(2);
    bool loose = false;
    {
        const uint32_t lane = threadIdx.x & 63u;
        unsigned long long rem = __ballot(sampled);
        while (rem) {  // (wave-uniform: once per path that has lanes in this wave)
            const uint32_t l0 = (uint32_t)__ffsll((long long)rem) - 1u;
            const uint32_t kk = (uint32_t)__shfl((int)k, (int)l0);
            const unsigned long long same = __ballot(sampled && k == kk), jum = __ballot(sampled && k == kk && jumbled);
            const bool lw = (uint32_t)__builtin_popcountll(jum) * 2u > (uint32_t)__builtin_popcountll(same);
            if (lw && k == kk) loose = true;
            if (lw && lane == l0) {
                const uint32_t g = ord_group[kk];
                if (atomicOr(group_loose + g, 1u) == 0u) {  // the first to flag the group enters it into the list, and flags its entries
                    const uint32_t li = atomicAdd(probe_stats + 2, 1u);
                    if (li < LOOSE_MAX) probe_stats[4u + li] = g;
                    uint32_t k0 = kk, k1 = kk + 1u;
                    while (k0 > 0u && ord_group[k0 - 1u] == g) --k0;
                    while (k1 < n_ordered && ord_group[k1] == g) ++k1;
                    uint64_t vol = 0;  // the steps of the group: what the tail kernel will have to mark, one atomic each
                    for (uint32_t q = k0; q < k1; ++q) {
                        entry_loose[q] = 1u;
                        vol += ent_len[q];
                    }
                    atomicAdd(probe_stats + 3, (uint32_t)((vol + 1023u) >> 10));
                }
            }
            rem &= ~same;
        }
    }
    stamp(3);
    if (len && loose) {
        const uint64_t pr = (uint64_t)e * len / n_bands;  // (never read: the coverage kernel leaves the group out)
        desc = end_a > end_z;
        j = desc ? len - pr : pr;
    } else if (len) {
        const uint32_t a = end_a, z = end_z;
        desc = a > z;
        uint32_t ka_fix = 0xFFFFFFFFu, kz_fix = 0u;  // (keys of the inner samples: smallest, largest)
        if (len >= 16) {
            const uint32_t q1 = in_q1, q2 = in_q2, q3 = in_q3;
            const int down = (int)(a > q1) + (int)(q1 > q2) + (int)(q2 > q3) + (int)(q3 > z);
            const int up = (int)(a < q1) + (int)(q1 < q2) + (int)(q2 < q3) + (int)(q3 < z);
            if (down != up) desc = down > up;
            const uint32_t k1 = desc ? ~q1 : q1, k2 = desc ? ~q2 : q2, k3 = desc ? ~q3 : q3;
            ka_fix = k1 < k2 ? (k1 < k3 ? k1 : k3) : (k2 < k3 ? k2 : k3);
            kz_fix = k1 > k2 ? (k1 > k3 ? k1 : k3) : (k2 > k3 ? k2 : k3);
        }
        // the ends bracket the search; an end that is not where the rest of the path says (a first step from elsewhere, the path's
        // start visited again at its end) would send every edge to that end, and with it every step out of its band: such an end is
        // taken for "below / above everything" instead, and the search finds the edges inside
        uint32_t ka = desc ? ~a : a, kz = desc ? ~z : z;
        if (ka > ka_fix) ka = 0u;
        if (kz < kz_fix) kz = 0xFFFFFFFFu;
        if (e == 0) j = desc ? len : 0;
        else if (e == n_bands) j = desc ? 0 : len;
        else {
            const uint32_t x = e * band_items;  // 1 <= x <= n_items: inner edges only
            j = desc ? band_edge_search<true>(items, ps, len, ~(x - 1u), ka, kz, n_probes, n_astray)
                     : band_edge_search<false>(items, ps, len, x, ka, kz, n_probes, n_astray);
        }
    }
    if (dbg_wave && j == 0x7FFFFFFFFFFFFFFFull) dbg_time[15] = 3;
    stamp(4);
    if (dbg_wave) {
        uint32_t np = n_probes;
        for (int o = 32; o > 0; o >>= 1) np += __shfl_down(np, o);
        if (threadIdx.x == 0) dbg_time[8] = np;
    }
    // How the probes fared, summed over the kernel: a sector with a step that cannot stand between the ends of the bracket it
    // was probed in is a sign of LONG-RANGE disorder (local jitter stays between the ends).  Where a quarter of the probes met
    // one, the paths do not follow the ids and the coverage kernel does not start: k_band_cover reads the two sums.
    {
        __shared__ uint32_t s_np, s_na;
        if (threadIdx.x == 0) s_np = s_na = 0;
        __syncthreads();
        uint32_t np = n_probes, na = n_astray;
        for (int o = 32; o > 0; o >>= 1) {
            np += __shfl_down(np, o);
            na += __shfl_down(na, o);
        }
        // (a sample: every 16th workgroup reports -- thousands of atomics on two words would outlast the searches; a grid of a few
        // hundred workgroups reports in full: sixteen of them would speak for two or three entries, and ONE piece of a path that the
        // searches have trouble with would be taken for the whole graph)
        const bool reports = (blockIdx.x & 15u) == 0u || gridDim.x <= 256u;
        if (reports && (threadIdx.x & 63u) == 0 && np) {
            atomicAdd(&s_np, np);
            if (na) atomicAdd(&s_na, na);
        }
        __syncthreads();
        if (reports && threadIdx.x == 0 && s_np) {
            atomicAdd(probe_stats, s_np);
            if (s_na) atomicAdd(probe_stats + 1, s_na);
        }
    }
    stamp(5);
    // A search that probed a stretch of the path that is out of place (a translocated block, a copy of another region) may end
    // far from the crossing; the segment between it and its neighbour would then hold thousands of steps of other bands -- all
    // spilled, and one wave of the coverage kernel streaming them while its workgroup waits.  Edge positions of a path run
    // one way, so a position that its neighbours (the lanes next to this one: consecutive edges of the same path) do not
    // bracket is replaced by their median: of five where two neighbours exist on either side, of three, or held against the
    // one neighbour there is.  Any positions make a valid cover of the path; these make one without outliers.
    {
        const uint32_t lane = threadIdx.x & 63u;
        auto nb = [&](int d, uint64_t &v) {  // the position of edge e + d of the same path, if a lane of this wave holds it
            const int src = (int)lane + d;
            const uint64_t pv = __shfl(j, src & 63);
            const uint32_t kv = __shfl(k, src & 63);
            v = pv;
            return src >= 0 && src < 64 && kv == k && live;
        };
        uint64_t m2, m1, p1, p2;
        const bool hm2 = nb(-2, m2), hm1 = nb(-1, m1), hp1 = nb(1, p1), hp2 = nb(2, p2);
        auto lo2 = [](uint64_t &x, uint64_t &y) {
            if (y < x) {
                const uint64_t t = x;
                x = y;
                y = t;
            }
        };
        if (live && len && e != 0 && e != n_bands) {
            if (hm2 && hm1 && hp1 && hp2) {
                uint64_t v0 = m2, v1 = m1, v2 = j, v3 = p1, v4 = p2;  // median of five
                lo2(v0, v1), lo2(v3, v4), lo2(v0, v3), lo2(v1, v4), lo2(v1, v2), lo2(v2, v3), lo2(v1, v2);
                j = v2;
            } else if (hm1 && hp1) {
                uint64_t v0 = m1, v1 = j, v2 = p1;
                lo2(v0, v1), lo2(v1, v2), lo2(v0, v1);
                j = v1;
            } else if (hp1) {
                if (desc ? j < p1 : j > p1) j = p1;
            } else if (hm1) {
                if (desc ? j > m1 : j < m1) j = m1;
            }
        }
    }
    if (live) bidx[tid] = (ps + j) | (desc ? BAND_DESC : 0ull);
    stamp(6);
}

// Graphs of many short paths (the contigs of an assembly-based pangenome: thousands of paths that each touch a few bands): most
// (entry, band) cells are empty, and a workgroup that takes the visiting order four entries at a time pays a barrier and a
// fold for every four empty cells (4 M items x 4000 paths: 0.67 ms for 1.2 GB of steps).  For such shapes the entries that DO
// have steps on a band are listed first -- one wave per (band, split of the visiting order): the two edge positions of 64
// entries at a time, a ballot, the survivors written in order -- and k_band_cover<SPARSE> walks the list instead of the order.
// clist[band * n_ordered + k_lo(split) ...]: the entries, ccnt[split * n_bands + band]: how many.
__global__ __launch_bounds__(256) void k_band_compact(const unsigned long long *__restrict__ bidx, uint32_t n_bands, uint32_t n_ordered, BandSplits sp,
                                                      uint32_t *__restrict__ clist, uint32_t *__restrict__ ccnt) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t cell = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (cell >= n_bands * sp.n) return;
    const uint32_t band = cell % n_bands, split = cell / n_bands;
    const uint32_t k_lo = sp.k[split], k_hi = sp.k[split + 1];
    uint32_t *out = clist + (uint64_t)band * n_ordered + k_lo;
    uint32_t n = 0;
    for (uint32_t k0 = k_lo; k0 < k_hi; k0 += 64u) {
        const uint32_t k = k0 + lane;
        bool some = false;
        if (k < k_hi) {
            const unsigned long long *ek = bidx + (uint64_t)k * (n_bands + 1u) + band;
            some = (ek[0] & ~BAND_DESC) != (ek[1] & ~BAND_DESC);
        }
        const unsigned long long m = __ballot(some);
        if (some) out[n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = k;
        n += (uint32_t)__builtin_popcountll(m);
    }
    if (lane == 0) ccnt[cell] = n;
}

// The coverage kernel.  flags[6] += the steps found outside the band they were dealt to (spilled to sl.rec: k_band_tail);
// flags[5] |= 1 when an index entry is inconsistent: the result of the pass is void.
// SPLIT: the workgroups (band, split) of a band share its tiles: each adds its counters to the (zeroed) coverage vector and
// leaves the histogram to K2.
// SPARSE: the entries are those of the band's list (k_band_compact) instead of the visiting order itself.
template <int NPL, int CW, bool WRITE_M, int BAND_D, bool SPLIT, bool SPARSE>
__global__ __launch_bounds__(CW * 64) void k_band_cover(const uint32_t *__restrict__ items, const unsigned long long *__restrict__ bidx,
                                                        const uint32_t *__restrict__ ord_group, uint32_t n_ordered,
                                                        const uint8_t *__restrict__ exclude, uint32_t n_items, uint32_t n_tiles,
                                                        uint32_t *__restrict__ M, uint64_t row_words, uint32_t *__restrict__ countable,
                                                        RowHist hs, uint32_t *__restrict__ flags, uint32_t n_bands, BandSplits sp,
                                                        BandSpill sl, const uint32_t *__restrict__ probe_stats,
                                                        const uint32_t *__restrict__ clist, const uint32_t *__restrict__ ccnt,
                                                        const uint32_t *__restrict__ entry_loose) {
    constexpr int BT = CW;  // tiles per band = waves per workgroup
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t bm[2][CW][BT * 64];  // two generations of CW band bitmaps (one per entry of a batch)
    extern __shared__ unsigned long long sh_hist[];

    // the index kernel's probes say the paths do not follow the ids (a quarter of the sectors it looked at held steps from
    // elsewhere): nearly every step would be spilled and the pass void at the end -- it is void now, and nothing is read
    if (probe_stats[1] > 16u && (unsigned long long)probe_stats[1] * 4ull > probe_stats[0]) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(flags + 5, 8u);
        return;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t bid = blockIdx.x;
    const uint32_t band = SPLIT ? bid % n_bands : bid;
    const uint32_t split = SPLIT ? bid / n_bands : 0u;
    // this workgroup's entries: a range of the visiting order -- or (SPARSE) of the band's list of entries that have steps here
    const uint32_t k_lo = SPARSE ? 0u : (SPLIT ? sp.k[split] : 0u);
    const uint32_t k_hi = SPARSE ? ccnt[bid] : (SPLIT ? sp.k[split + 1] : n_ordered);
    const uint32_t *mine_list = SPARSE ? clist + (uint64_t)band * n_ordered + (SPLIT ? sp.k[split] : 0u) : nullptr;
    const uint32_t tile = band * BT + wave;
    const bool active = tile < n_tiles;  // the last band may hold fewer tiles; its spare waves still stream segments
    for (uint32_t i = threadIdx.x; i < 2 * CW * BT * 64; i += CW * 64) (&bm[0][0][0])[i] = 0;
    if (!SPLIT && hs.rep)
        for (uint32_t b = threadIdx.x; b <= hs.n_groups; b += CW * 64) sh_hist[b] = 0;
    __syncthreads();
    const uint32_t excl = active ? tile_exclusion_word(exclude, tile, lane, n_items) : 0u;
    // ids of this band: [lo_id, lo_id + width); item 0 is no item, ids beyond n_items are none either
    const uint32_t lo_id = band ? band * (uint32_t)(BT * BLOCK_ITEMS) : 1u;
    const uint64_t hi64 = (uint64_t)(band + 1) * (BT * BLOCK_ITEMS);
    const uint32_t hi_id = hi64 > (uint64_t)n_items + 1 ? n_items + 1u : (uint32_t)hi64;
    const uint32_t width = hi_id - lo_id;

    TileCounters<NPL> tc;
    uint32_t acc = 0, cur_g = NONE;
    bool bad = false;
    auto flush = [&](uint32_t g) {
        const uint32_t x = acc & ~excl;
        acc = 0;
        if (WRITE_M && active) M[(uint64_t)g * row_words + (uint64_t)tile * BLOCK_WORDS + lane] = x;
        tc.add_group(x);
    };

    // ---- 64 entries of the visiting order in the lanes: their segments on this band (issue side) ----
    uint64_t w_lo = 0;
    uint32_t w_len = 0, swin = NONE;  // swin: which window of 64 entries
    uint32_t w_g = NONE;              // ... and their groups: handed to the fold side when it reaches the window
    const uint32_t n_edges = n_bands + 1;  // per entry: n_bands + 1 edge positions
    auto load_swin = [&](uint32_t win) {
        swin = win;
        const uint32_t kc = win * 64u + lane;
        uint64_t a = 0, b = 0;
        w_g = NONE;
        if (kc >= k_lo && kc < k_hi) {
            const uint32_t k = SPARSE ? mine_list[kc] : kc;
            const unsigned long long *ek = bidx + (uint64_t)k * n_edges + band;
            const uint32_t lz = entry_loose[k];
            a = ek[0];
            b = ek[1];
            w_g = ord_group[k];
            if (lz != 0u) a = b = 0;  // an entry of a group left to the bitmaps (BandLoose): no steps of it here
        }
        if (((a ^ b) & BAND_DESC) != 0) bad = true;
        a &= ~BAND_DESC;
        b &= ~BAND_DESC;
        // the interval between the two edge positions, whichever comes first: the intervals of a path cover all its steps
        // even where the searches of a path that is not sorted returned positions out of order
        const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
        uint64_t len = hi - lo;
        if (len >= (1ull << 29)) {  // (a buffer descriptor holds 2^32 bytes)
            bad = true;
            len = 0;
        }
        w_lo = lo;
        w_len = (uint32_t)len;
    };
    // ---- the groups of the window being folded (the issue side may already be one window ahead; it has loaded this
    // window before the first fold in it, and moves on only after that fold: no load on the fold side, whose wait would
    // drain the next segment's loads in flight) ----
    uint32_t f_g = NONE, fwin = NONE;

    // A segment is read through a buffer descriptor (base = its 16-byte aligned start, size = its bytes): loads beyond its
    // end return zeros without a branch, so every load is issued unconditionally and the compiler counts them exactly
    // (s_waitcnt vmcnt(n) per slot instead of a drain in front of every use).
    struct Seg {
        __amdgpu_buffer_rsrc_t rs;
        const uint32_t *base;
        uint32_t head, nal;  // steps of the first load before the segment; head + length
        uint32_t g;          // the group of the segment's entry (wave-uniform): what a spilled step is recorded under
    };
    auto make_seg = [&](uint64_t lo, uint32_t len, uint32_t g) {
        const uint32_t *base = items + (lo & ~3ull);
        const uint32_t head = len ? (uint32_t)(lo & 3ull) : 0u, nal = head + len;
        const uint32_t b_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)base);
        const uint32_t b_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)base >> 32));
        void *bp = (void *)(((uintptr_t)b_hi << 32) | b_lo);
        return Seg{__builtin_amdgcn_make_buffer_rsrc(bp, 0, (int)__builtin_amdgcn_readfirstlane(nal * 4u), 0x00020000), (const uint32_t *)bp, head, nal, g};
    };
    auto seg_of = [&](uint32_t batch) {
        const uint32_t k = batch * CW + wave;
        if ((k >> 6) != swin) load_swin(k >> 6);  // (every wave: the fold needs the window's groups whether or not this wave has an entry)
        if (k < k_lo || k >= k_hi) return make_seg(0, 0, NONE);
        const uint32_t l = k & 63u;
        const uint32_t lo_l = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w_lo, l);
        const uint32_t lo_h = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w_lo >> 32), l);
        const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)w_len, l);
        return make_seg(((uint64_t)lo_h << 32) | lo_l, len, (uint32_t)__builtin_amdgcn_readlane((int)w_g, l));
    };

    // batches of CW entries, numbered over the whole visiting order (64 is a multiple of CW: a batch never straddles two
    // windows); a split's first and last batch may hold entries of its neighbours: they are empty here
    const uint32_t batch_lo = k_lo / CW, batch_hi = (k_hi + CW - 1) / CW;
    auto issue = [&](const Seg &s, uint32_t r0, int u) {
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(s.rs, lane * 16u + (uint32_t)u * 1024u, r0 * 4u, /*nt*/ 2));
    };
    // the steps of one 16-byte load: presence bits into the band bitmap of the entry.  (Under the execution mask, not as an OR
    // of zero: the lanes past the end of a segment would all meet on one word, and same-address LDS atomics serialise.)
    auto process = [&](const u32x4 &v, const Seg &s, uint32_t r0, int u, uint32_t *map) {
        const uint32_t q = r0 + (uint32_t)u * 256u + lane * 4u - s.head;  // position in the segment (wraps before it)
        const uint32_t len = s.nal - s.head;
        const uint32_t ids[4] = {v.x, v.y, v.z, v.w};
        uint32_t out = 0;  // which of the four steps lie outside the band
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t id = ids[e];
            const bool valid = q + (uint32_t)e < len;
            const bool inb = id - lo_id < width;
            if (valid && inb) atomicOr(&map[(((id >> 11) & (uint32_t)(BT - 1)) << 6) | (id & 63u)], 1u << ((id >> 6) & 31u));
            out |= (valid && !inb) ? 1u << e : 0u;
        }
        if (__builtin_expect(__ballot(out != 0) != 0ull, 0)) {
            // spilled: one slot range of the list per wave and load (a path that leaves its band does so for a stretch of steps)
            const uint32_t n_mine = (uint32_t)__builtin_popcount(out);
            uint32_t incl = n_mine;  // inclusive prefix sum over the lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o);
                if (lane >= (uint32_t)o) incl += t;
            }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            unsigned long long took = 0;
            if (lane == 0) took = atomicAdd(reinterpret_cast<unsigned long long *>(flags + 6), (unsigned long long)total | (1ull << 32));
            const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)took);
            const uint32_t burst = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(took >> 32));
            if (lane == 0 && burst < sl.dir_cap) sl.dir[burst] = ((unsigned long long)base << 32) | s.g;
            uint32_t at = base + incl - n_mine;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if ((out >> e) & 1u) {
                    if (at < sl.cap) sl.rec[at] = ids[e];
                    ++at;
                }
            }
        }
    };
    auto fold = [&](uint32_t batch) {
        const uint32_t k0 = batch * CW;
        if ((k0 >> 6) != fwin) {  // the first batch of a window: w_g still holds this window's groups
            fwin = k0 >> 6;
            f_g = w_g;
        }
        uint32_t x[CW];
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            uint32_t *w = &bm[batch & 1u][i][wave * 64u + lane];
            x[i] = *w;
            *w = 0;
        }
#pragma unroll
        for (int i = 0; i < CW; ++i) {
            const uint32_t k = k0 + (uint32_t)i;
            if (k >= k_lo && k < k_hi) {
                const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)f_g, k & 63u);
                if (g != cur_g) {
                    if (cur_g != NONE) flush(cur_g);
                    cur_g = g;
                }
                acc |= x[i];
            }
        }
    };

    if (batch_lo < batch_hi) {
        Seg cur = seg_of(batch_lo);
        // (the groups of the first window, taken now: where a split begins with the LAST batch of a window the issue side has
        // loaded the next window by the time that batch is folded)
        fwin = (batch_lo * CW) >> 6;
        f_g = w_g;
        uint32_t c_r0 = 0, c_batch = batch_lo;
        // one group of BAND_D loads per lane: consume `in` slot by slot, the next group's loads going out into `out` as the
        // slots are taken -- BAND_D loads in flight throughout.  (Two register sets that swap roles: with one set refilled
        // in place the compiler parks the new loads elsewhere and copies them back at the loop head, i.e. waits for them.)
        auto group = [&](const u32x4(&in)[BAND_D], u32x4(&out)[BAND_D]) {
            Seg nxt = cur;
            uint32_t n_r0 = c_r0 + 256u * BAND_D, n_batch = c_batch;
            if (n_r0 >= cur.nal) {  // the segment ends with this group of loads
                n_batch = c_batch + 1;
                n_r0 = 0;
                nxt = n_batch < batch_hi ? seg_of(n_batch) : make_seg(0, 0, NONE);
            }
            uint32_t *map = &bm[c_batch & 1u][wave][0];
#pragma unroll
            for (int u = 0; u < BAND_D; ++u) {
                out[u] = issue(nxt, n_r0, u);
                if (c_r0 + (uint32_t)u * 256u < cur.nal) process(in[u], cur, c_r0, u, map);  // (wave-uniform: the slot holds steps)
                __builtin_amdgcn_sched_barrier(0);  // slot by slot: the other slots' loads stay in flight meanwhile
            }
            if (n_batch != c_batch) {
                __syncthreads();
                fold(c_batch);
            }
            cur = nxt;
            c_r0 = n_r0;
            c_batch = n_batch;
        };
        u32x4 bufA[BAND_D], bufB[BAND_D];
#pragma unroll
        for (int u = 0; u < BAND_D; ++u) bufA[u] = issue(cur, 0, u);
        while (true) {
            group(bufA, bufB);
            if (c_batch >= batch_hi) break;
            group(bufB, bufA);
            if (c_batch >= batch_hi) break;
        }
        if (cur_g != NONE) flush(cur_g);
    }
    tc.settle();
    if (__ballot(bad) && lane == 0) atomicOr(flags + 5, 1u);
    if (SPLIT) {
        // the band's other workgroups hold the counts of the other groups: they meet in the coverage vector
        if (active) {
            for (uint32_t b = 0; b < 32; ++b) {
                uint32_t v = 0;
#pragma unroll
                for (int k = 0; k < NPL; ++k) v |= ((tc.cnt[k] >> b) & 1u) << k;
                const uint64_t node = (uint64_t)tile * BLOCK_ITEMS + b * 64u + lane;
                if (node == 0) {
                    if (split == 0) countable[0] = 0xFFFFFFFFu;  // the reference's reserved element (abacus.rs:549-551)
                } else if (node <= n_items && v) {
                    atomicAdd(countable + node, v);
                }
            }
        }
        return;
    }
    if (active) tile_tail<NPL>(tc.cnt, tile, lane, n_items, countable, hs, sh_hist, 0u, 32u, 0xFFFFFFFFu);
    if (hs.rep) {
        __syncthreads();
        hist_bins_flush(hs, sh_hist, CW * 64);
    }
}

// ------------------------------------------------------------------------------------------
// the tail of a one-shot pass: spilled steps, then the histogram handed over
// ------------------------------------------------------------------------------------------
struct BandTail {
    const uint32_t *items;
    const unsigned long long *bidx;
    const uint32_t *group_first;  // n_groups + 1: the entries of group g are [group_first[g], group_first[g + 1])
    const uint8_t *exclude;
    uint32_t n_edges, n_items;
    BandSpill sl;
    unsigned long long *hset;  // (generation << 56) | (group << 32) | id: the pairs this pass has added
    uint32_t hmask, gen;
    uint32_t *countable;
    uint32_t *M;  // nullptr: the pass writes no presence matrix
    uint64_t row_words;
    RowHist hs;                // rep == nullptr: K2 takes the histogram from the coverage vector afterwards
    unsigned long long *hist;  // the pass's histogram in HBM
    uint32_t *flags;
    uint32_t *scratch;         // (cleared with the pass's counters) [0] arrivals of the workgroups, [1] steps the scans have read, in units of 1024
    uint32_t *probe_stats;     // the index kernel's two sums (probes, probes that met steps astray): set back to zero for the next pass
    uint32_t *host_block;      // [flags u32[8] | hist] in the ticket's pinned memory, or nullptr
    uint32_t scan_budget;      // steps the spill scans may read in all, in units of 1024
    BandLoose lo;              // the groups that were left to bitmaps, and the bitmaps
    const uint64_t *ent_start;  // ... their entries' steps
    const uint64_t *ent_len;
    uint32_t n_tiles, n_groups, n_ordered;
    uint32_t loose_budget;     // steps the loose groups may hold in all, in units of 1024
};

typedef __attribute__((address_space(1))) uint32_t g_u32;
typedef __attribute__((address_space(1))) unsigned long long g_u64;
__device__ static inline uint32_t agent_load(const uint32_t *p) {
    return __hip_atomic_load((g_u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ static inline unsigned long long agent_load(const unsigned long long *p) {
    return __hip_atomic_load((g_u64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// true: (g, id) was not in the set of this pass.  Slots of older passes (another generation) count as empty; the table is
// at most half full (2 x the capacity of the spill list), so a probe sequence ends.  The compare-and-swap decides: a load
// that returned an outdated slot costs one more round, never a wrong answer (a slot changes once per pass).
__device__ static inline bool spill_set_insert(unsigned long long *slots, uint32_t mask, uint32_t gen, uint32_t g, uint32_t id) {
    const unsigned long long key = ((unsigned long long)gen << 56) | ((unsigned long long)g << 32) | id;
    unsigned long long z = key * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 29)) * 0xBF58476D1CE4E5B9ull;
    uint32_t h = (uint32_t)(z >> 32) & mask;
    for (;;) {
        const unsigned long long cur = agent_load(slots + h);
        if (cur == key) return false;
        if ((uint32_t)(cur >> 56) != gen) {
            const unsigned long long old = atomicCAS(slots + h, cur, key);
            if (old == cur) return true;
            if (old == key) return false;
            if ((uint32_t)(old >> 56) != gen) continue;
        }
        h = (h + 1u) & mask;
    }
}

// The spill list, one wave per burst.  Every word that several workgroups touch -- flags, the set, the coverage vector, the
// histogram replicas, M -- is touched by device-scope atomics and agent-scope loads only (the L2s of the XCDs are not
// coherent with each other); the workgroup that arrives last adds the replicas up.
// What it costs is the chain of dependent reads a burst starts at places nobody has touched before (its records -> the band
// edges of its group's entries -> the segment; then set slot -> coverage word): ~7 us per burst and wave on 10 M x 256
// whatever the volume of the scans (DESIGN.md, K-band), so bursts are what the time scales with, 8192 of them at a time.
// (The histogram bins an added pair moves are the workgroup's LDS bins, flushed once at the end: most pairs move an item from
// bin 0 to bin 1 or from 1 to 2, and two atomics per pair on the same few words of a replica in HBM are performed one after the
// other, ~12 ns each -- 1 M pairs on 64 replicas: 0.19 of the tail kernel's 0.21 ms on pansyn-v1r 10 M x 256.)
struct TailAdd {
    const BandTail &a;
    unsigned long long *bins;  // the workgroup's histogram bins in LDS (hs.rep != nullptr)
    __device__ __forceinline__ void operator()(uint32_t g, uint32_t id) const {
        if (a.exclude && a.exclude[id]) return;
        if (!spill_set_insert(a.hset, a.hmask, a.gen, g, id)) return;
        const uint32_t old = atomicAdd(a.countable + id, 1u);  // AbacusByTotal::coverage: one more group visits the item
        if (a.hs.rep && old < a.hs.n_groups) {
            const unsigned long long w = a.hs.weights ? (unsigned long long)a.hs.weights[id] : 1ull;
            atomicAdd(&bins[old], 0ull - w);
            atomicAdd(&bins[old + 1u], w);
        }
        if (a.M) atomicOr(a.M + (uint64_t)g * a.row_words + (uint64_t)(id >> 11) * BLOCK_WORDS + (id & 63u), 1u << ((id >> 6) & 31u));
    }
};

__global__ __launch_bounds__(BAND_TAIL_WAVES * 64) void k_band_tail(BandTail a) {
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t bmp_all[BAND_TAIL_WAVES][256];  // per wave: the 8192 ids of one band
    __shared__ uint32_t s_last;
    extern __shared__ unsigned long long t_hist[];  // n_groups + 1 bins (a pass that adds its histogram itself): what the bitmaps move
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *bmp = bmp_all[wave];
    const TailAdd add{a, t_hist};
    // ---- the groups that were left to bitmaps (BandLoose), by the first LOOSE_WORKGROUPS workgroups of the grid (they are
    // dispatched first, so all of them get to run whatever else is on the chip: they may wait for each other).  First the
    // groups' steps, wherever they lie, become presence bits of the group's bitmap; then -- all marks made -- a wave takes a
    // tile at a time and adds every such group's presence word to the tile's items: coverage word + 1 (atomically: the spilled
    // steps below add to the same vector), the item's weight moved from bin `old` to bin `old + 1` in the workgroup's LDS
    // bins; it writes the group's row of the presence matrix and clears the bitmap behind it.  (state[2]: how many groups;
    // beyond LOOSE_MAX nothing is marked and the pass is void: the path rows serve any path.)
    const uint32_t nl_raw = a.lo.state[2];
    const bool loose_ok = nl_raw <= LOOSE_MAX && a.lo.state[3] <= a.loose_budget;  // (state[3]: their steps, in units of 1024)
    const uint32_t nl = loose_ok ? nl_raw : 0u;
    if (!loose_ok && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.flags + 5, 16u);
    // the workgroup's histogram bins: what the bitmaps and the spilled steps move, flushed once before the workgroup arrives
    const bool lds_bins = a.hs.rep && (nl != 0u || agent_load(a.flags + 7) != 0u);
    if (lds_bins) {
        for (uint32_t b = threadIdx.x; b <= a.n_groups; b += blockDim.x) t_hist[b] = 0;
        __syncthreads();
    }
    if (nl && blockIdx.x < LOOSE_WORKGROUPS) {
        const uint64_t wid = (uint64_t)blockIdx.x * BAND_TAIL_WAVES + wave, n_waves = (uint64_t)LOOSE_WORKGROUPS * BAND_TAIL_WAVES;
        bool bad_id = false;
        for (uint32_t li = 0; li < nl; ++li) {
            const uint32_t g = a.lo.state[4u + li];
            uint32_t *bits = a.lo.bits + (uint64_t)li * a.n_tiles * BLOCK_WORDS;
            for (uint32_t k = a.group_first[g]; k < a.group_first[g + 1]; ++k) {
                const uint64_t ps = a.ent_start[k], pe = ps + a.ent_len[k];
                for (uint64_t c = (ps & ~3ull) + wid * 1024u; c < pe; c += n_waves * 1024u) {
                    u32x4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint64_t q = c + (uint64_t)u * 256u + lane * 4u;
                        v[u] = q < pe ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.items + q)) : u32x4{0, 0, 0, 0};
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint64_t q = c + (uint64_t)u * 256u + lane * 4u;
                        const uint32_t ids[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t id = ids[e];
                            if (q + (uint32_t)e >= ps && q + (uint32_t)e < pe) {
                                if (id - 1u < a.n_items)
                                    __hip_atomic_fetch_or(bits + (uint64_t)(id >> 11) * BLOCK_WORDS + (id & 63u), 1u << ((id >> 6) & 31u), __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
                                else bad_id = true;  // (an id that is no item: the rows route reports it)
                            }
                        }
                    }
                }
            }
        }
        if (__ballot(bad_id) && lane == 0) atomicOr(a.flags + 5, 32u);
        // all marks of all the marking workgroups made (device-scope atomics, acknowledged) before any bitmap is read
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add((g_u32 *)(a.scratch + 2), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (a poll every ~3 us: the arrivals queue on the same word.  Bounded: the 256 workgroups are the first of the grid and the
            // chip holds three times as many, but should several contexts' tail kernels ever share it so that none has all its
            // markers resident, the wait ends after ~70 ms: the pass is void -- flags[5] |= 64 --, the host clears the bitmaps)
            uint32_t polls = 0;
            while (agent_load(a.scratch + 2) < LOOSE_WORKGROUPS && polls < 20000u) {
                __builtin_amdgcn_s_sleep(127);
                ++polls;
            }
            if (polls >= 20000u) atomicOr(a.flags + 5, 64u);
        }
        __syncthreads();
        for (uint32_t tile = blockIdx.x * BAND_TAIL_WAVES + wave; tile < a.n_tiles; tile += LOOSE_WORKGROUPS * BAND_TAIL_WAVES) {
            uint32_t excl = 0xFFFFFFFFu;  // (computed for the first bitmap that holds anything on the tile)
            for (uint32_t li = 0; li < nl; ++li) {
                uint32_t *word = a.lo.bits + ((uint64_t)li * a.n_tiles + tile) * BLOCK_WORDS + lane;
                uint32_t x = agent_load(word);
                if (__ballot(x != 0u) == 0ull) continue;
                if (x) *word = 0;
                if (excl == 0xFFFFFFFFu) excl = tile_exclusion_word(a.exclude, tile, lane, a.n_items);
                x &= ~excl;
                const uint32_t g = a.lo.state[4u + li];
                if (a.M) a.M[(uint64_t)g * a.row_words + (uint64_t)tile * BLOCK_WORDS + lane] = x;
                // (eight coverage words at a time: their atomics are on their way together, the bins follow when the old values are back)
                for (uint32_t b0 = 0; b0 < 32; b0 += 8) {
                    if (__ballot(((x >> b0) & 0xFFu) != 0u) == 0ull) continue;
                    uint32_t old[8];
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j) {
                        const uint32_t id = tile * BLOCK_ITEMS + (b0 + j) * 64u + lane;
                        old[j] = ((x >> (b0 + j)) & 1u) ? atomicAdd(a.countable + id, 1u) : 0xFFFFFFFFu;
                    }
                    if (a.hs.rep) {
#pragma unroll
                        for (uint32_t j = 0; j < 8; ++j) {
                            if (old[j] < a.n_groups) {
                                const uint32_t id = tile * BLOCK_ITEMS + (b0 + j) * 64u + lane;
                                const unsigned long long w = a.hs.weights ? (unsigned long long)a.hs.weights[id] : 1ull;
                                atomicAdd(&t_hist[old[j]], 0ull - w);
                                atomicAdd(&t_hist[old[j] + 1u], w);
                            }
                        }
                    }
                }
            }
        }
    }
    uint32_t n = agent_load(a.flags + 6), n_bursts = agent_load(a.flags + 7);
    if (n > a.sl.cap || n_bursts > a.sl.dir_cap) {  // the list did not hold them all: the pass is void
        if (threadIdx.x == 0) atomicOr(a.flags + 5, 2u);
        n = n_bursts = 0;
    }
    bool over = false;
    uint32_t vol_acc = 0;
    const uint32_t bu_step = gridDim.x * BAND_TAIL_WAVES;
    uint32_t bu = blockIdx.x * BAND_TAIL_WAVES + wave;
    // (the directory entries of the next burst are asked for while this one is scanned: one link less in the chain)
    unsigned long long d0 = bu < n_bursts ? a.sl.dir[bu] : 0ull, d1 = bu + 1u < n_bursts ? a.sl.dir[bu + 1u] : 0ull;
    for (; bu < n_bursts && !over; bu += bu_step) {
        const uint32_t start = (uint32_t)(d0 >> 32), lg = (uint32_t)d0;
        uint32_t end = bu + 1u < n_bursts ? (uint32_t)(d1 >> 32) : n;
        if (end > n || end < start || end - start > 256u) end = start;  // (cannot be: a burst is one load of one wave)
        {
            const uint32_t nb = bu + bu_step;
            d0 = nb < n_bursts ? a.sl.dir[nb] : 0ull;
            d1 = nb + 1u < n_bursts ? a.sl.dir[nb + 1u] : 0ull;
        }
        uint32_t idv[4], pend = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t i = start + (uint32_t)r * 64u + lane;
            idv[r] = i < end ? a.sl.rec[i] : 0u;
            pend |= (i < end && idv[r] >= 1u && idv[r] <= a.n_items) ? 1u << r : 0u;
        }
        const uint32_t k0 = a.group_first[lg], k1 = a.group_first[lg + 1];
        while (!over) {
            // the band of the first record still pending: the burst's records on that band are one cell (lg, lb)
            uint32_t lb = NONE;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned long long m = __ballot(((pend >> r) & 1u) != 0);
                if (lb == NONE && m) lb = (uint32_t)__builtin_amdgcn_readlane((int)(idv[r] >> BAND_SHIFT), __ffsll((long long)m) - 1);
            }
            if (lb == NONE) break;
            uint32_t mine = 0, mn = 0xFFFFFFFFu, mx = 0u;  // ... and the range of their ids: what a scanned step is first held against
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (((pend >> r) & 1u) && (idv[r] >> BAND_SHIFT) == lb) {
                    mine |= 1u << r;
                    mn = idv[r] < mn ? idv[r] : mn;
                    mx = idv[r] > mx ? idv[r] : mx;
                }
            }
            pend &= ~mine;
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t t0 = __shfl_xor(mn, o), t1 = __shfl_xor(mx, o);
                mn = t0 < mn ? t0 : mn;
                mx = t1 > mx ? t1 : mx;
            }
            mn = (uint32_t)__builtin_amdgcn_readfirstlane(mn);
            const uint32_t span = (uint32_t)__builtin_amdgcn_readfirstlane(mx) - mn;
            // the records of the cell as a bitmap of the band
#pragma unroll
            for (int w = 0; w < 4; ++w) bmp[w * 64 + lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((mine >> r) & 1u) atomicOr(&bmp[(idv[r] & ((1u << BAND_SHIFT) - 1u)) >> 5], 1u << (idv[r] & 31u));
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // what the group's paths visit IN the band was counted by the coverage kernel: clear it
            uint32_t vol = 0;
            for (uint32_t k = k0; k < k1; ++k) {
                const unsigned long long *ek = a.bidx + (uint64_t)k * a.n_edges + lb;
                const uint64_t e0 = ek[0] & ~BAND_DESC, e1 = ek[1] & ~BAND_DESC;
                const uint64_t lo = e0 < e1 ? e0 : e1;
                const uint32_t seg_len = (uint32_t)((e0 < e1 ? e1 : e0) - lo);
                vol += (seg_len + 1023u) >> 10;
                // (SCAN_U 16-byte loads per lane in flight; positions relative to the aligned start of the segment: a segment holds
                // fewer than 2^29 steps)
                const u32x4 *src = reinterpret_cast<const u32x4 *>(a.items + (lo & ~3ull));
                const uint32_t head = (uint32_t)(lo & 3ull), nal = head + seg_len;  // steps before the segment in its first load; + its length
                for (uint32_t r0 = 0; r0 < nal; r0 += 256u * SCAN_U) {
                    u32x4 v[SCAN_U];
#pragma unroll
                    for (int u = 0; u < SCAN_U; ++u) {
                        const uint32_t q = r0 + (uint32_t)u * 256u + lane * 4u;
                        v[u] = q < nal ? __builtin_nontemporal_load(src + (q >> 2)) : u32x4{0, 0, 0, 0};
                    }
#pragma unroll
                    for (int u = 0; u < SCAN_U; ++u) {
                        const uint32_t q = r0 + (uint32_t)u * 256u + lane * 4u;
                        const uint32_t ids[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                        uint32_t hit = 0;  // steps of the segment with an id in the range of the records
#pragma unroll
                        for (int e = 0; e < 4; ++e) hit |= (ids[e] - mn <= span && q + (uint32_t)e - head < seg_len) ? 1u << e : 0u;
                        if (__ballot(hit != 0)) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if ((hit >> e) & 1u) {
                                    const uint32_t x = ids[e];
                                    const uint32_t xw = (x & ((1u << BAND_SHIFT) - 1u)) >> 5, xb = 1u << (x & 31u);
                                    if (bmp[xw] & xb) atomicAnd(&bmp[xw], ~xb);
                                }
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // what is left was visited only out of band: one lane per id adds it
            uint32_t wonm = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t id = idv[r];
                const uint32_t wi = (id & ((1u << BAND_SHIFT) - 1u)) >> 5, bit = 1u << (id & 31u);
                if (((mine >> r) & 1u) && (atomicAnd(&bmp[wi], ~bit) & bit) != 0) wonm |= 1u << r;
            }
            while (wonm) {  // (one copy of the code that adds a pair, not four)
                const int r = __builtin_ctz(wonm);
                wonm &= wonm - 1u;
                add(lg, r == 0 ? idv[0] : r == 1 ? idv[1] : r == 2 ? idv[2] : idv[3]);
            }
            __builtin_amdgcn_wave_barrier();
            // the scans are bounded: a graph whose paths do not follow the ids is served by path rows.  (The volume is taken to the
            // shared counter every 256 K steps a wave has read, not per cell: tens of thousands of atomics on one word would take
            // longer than the scans.)
            vol_acc += vol;
            if (vol_acc >= 256u) {
                uint32_t seen = 0;
                if (lane == 0) seen = atomicAdd(a.scratch + 1, vol_acc) + vol_acc;
                vol_acc = 0;
                seen = (uint32_t)__builtin_amdgcn_readfirstlane(seen);
                if (seen > a.scan_budget) {
                    if (lane == 0) atomicOr(a.flags + 5, 4u);
                    over = true;
                }
            }
        }
    }
    if (vol_acc && lane == 0 && atomicAdd(a.scratch + 1, vol_acc) + vol_acc > a.scan_budget) atomicOr(a.flags + 5, 4u);
    if (lds_bins) {
        __syncthreads();
        unsigned long long *rep = a.hs.rep + (size_t)(blockIdx.x % HIST_REPLICAS) * (a.n_groups + 1u);
        for (uint32_t b = threadIdx.x; b <= a.n_groups; b += blockDim.x) {
            const unsigned long long x = t_hist[b];
            if (x) atomicAdd(&rep[b], x);
        }
    }
    // ---- the histogram is handed over: by workgroup 0 when the list was empty (every workgroup knows: the count was final
    // when the kernel began), else by the last workgroup to arrive ----
    if (n_bursts || nl_raw) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // (what this workgroup wrote for others it wrote with device-scope atomics, which are performed where every XCD sees
        // them: once they are acknowledged -- the wait above -- the ticket may follow.  A release fence here would write back the
        // XCD's L2 once per workgroup, and those write-backs queue up: 0.1 ms for 2048 workgroups, measured)
        if (threadIdx.x == 0) {
            s_last = __hip_atomic_fetch_add((g_u32 *)a.scratch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
            if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (!s_last) return;
    } else {
        if (blockIdx.x != 0) return;
        __syncthreads();  // (the flag of a list that overflowed, set above)
    }
    if (a.hs.rep) {
        const uint32_t bins = a.hs.n_groups + 1u;
        for (uint32_t bn = threadIdx.x; bn < bins; bn += blockDim.x) {
            unsigned long long sum = 0;
            // (8 loads in flight at a time: all 64 at once are 128 registers, and the registers of this corner of the kernel
            // would set the occupancy of the spill scans)
            for (uint32_t r0 = 0; r0 < HIST_REPLICAS; r0 += 8) {
                unsigned long long v[8];
#pragma unroll
                for (uint32_t r = 0; r < 8; ++r) v[r] = agent_load(a.hs.rep + (size_t)(r0 + r) * bins + bn);
#pragma unroll
                for (uint32_t r = 0; r < 8; ++r) sum += v[r];
            }
            a.hist[bn] = sum;
            if (a.host_block) {
                a.host_block[8 + 2 * bn] = (uint32_t)sum;
                a.host_block[8 + 2 * bn + 1] = (uint32_t)(sum >> 32);
            }
        }
    }
    if (a.host_block && threadIdx.x < 8) a.host_block[threadIdx.x] = threadIdx.x == 3 ? nl_raw : agent_load(a.flags + threadIdx.x);
    if (!a.host_block && threadIdx.x == 0) a.flags[3] = nl_raw;  // (the pass's own copy of the block to the host follows this kernel)
    // the index kernel's block goes back to zero for the next pass: its statistics, the flags of the groups it found loose
    if (nl_raw > LOOSE_MAX) {
        for (uint32_t g = threadIdx.x; g < a.n_groups; g += blockDim.x) a.lo.group_flag[g] = 0;
        for (uint32_t k = threadIdx.x; k < a.n_ordered; k += blockDim.x) a.lo.entry_flag[k] = 0;
    } else {
        for (uint32_t li = 0; li < nl_raw; ++li) {
            const uint32_t g = a.lo.state[4u + li];
            for (uint32_t k = a.group_first[g] + threadIdx.x; k < a.group_first[g + 1]; k += blockDim.x) a.lo.entry_flag[k] = 0;
            if (threadIdx.x == 0) a.lo.group_flag[g] = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) a.probe_stats[threadIdx.x] = 0;
}

// The shapes the band route is worth it for: enough bands (times the splits of the visiting order) to fill the chip,
// segments long enough to stream, an index of reasonable size.  (Anything else -- and any graph whose paths stray from
// the order of the ids by more than the spill list holds -- takes the path rows.)
uint32_t band_route_splits(const pnx_ctx *ctx, uint32_t n_groups) {
    constexpr uint32_t BT = BAND_CW;
    const uint64_t n_bands = (ctx->n_blocks + BT - 1) / BT;
    if (const char *e = getenv("PNX_BAND_SPLITS")) {  // measurement: a fixed number of splits
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= BAND_MAX_SPLITS) return (uint32_t)std::min<uint64_t>((uint64_t)v, n_groups ? n_groups : 1);
    }
    // ~4.75 workgroups of 4 waves per CU is what the chip holds of this kernel
    const uint64_t want = (19ull * (uint64_t)ctx->prop.multiProcessorCount + 3) / 4;
    uint64_t s = n_bands ? (want + n_bands - 1) / n_bands : 1;
    if (n_bands >= 2ull * (uint64_t)ctx->prop.multiProcessorCount) s = 1;  // enough bands by themselves: one workgroup keeps a band's counters and adds the histogram itself
    s = std::min<uint64_t>(s, BAND_MAX_SPLITS);
    s = std::min<uint64_t>(s, n_groups ? n_groups : 1);
    return (uint32_t)(s ? s : 1);
}

bool band_route_fits(const pnx_ctx *ctx, uint32_t n_entries) {
    constexpr uint32_t BT = BAND_CW;
    if (!n_entries || !ctx->n_steps || !ctx->n_paths) return false;
    const uint64_t n_bands = (ctx->n_blocks + BT - 1) / BT;
    if (n_bands * BAND_MAX_SPLITS < 2ull * (uint64_t)ctx->prop.multiProcessorCount) return false;
    if (n_bands < 2ull * (uint64_t)ctx->prop.multiProcessorCount && n_entries < 2 * BAND_MAX_SPLITS) return false;  // (nothing to split)
    // segments long enough to stream: the paths must be long (a graph of many SHORT paths -- fewer than 4096 steps each on average --
    // has nothing but segment ends; one of many paths that each touch a few bands is fine: its empty cells are skipped, band_sparse)
    if (ctx->n_steps / ctx->n_paths < 4096) return false;
    if ((n_bands + 1) * n_entries * 12 > (384ull << 20)) return false;  // the index (8 bytes per cell) + the per-band entry lists (4)
    return true;
}

// the spill list and the set of added pairs: per context, made when the first one-shot pass is enqueued
static int ensure_spill(pnx_ctx *ctx) {
    uint64_t cap = ctx->n_steps / 64;
    cap = std::max<uint64_t>(cap, 1ull << 14);
    cap = std::min<uint64_t>(cap, 1ull << 25);
    uint64_t slots = 1;
    while (slots < 2 * cap) slots <<= 1;
    int rc;
    if (!ctx->d_band_probe.p) {  // the index kernel's probe statistics and its list of loose groups: zero once, every pass's tail sets them back
        if ((rc = ensure(ctx, ctx->d_band_probe, (4 + LOOSE_MAX) * 4))) return rc;
        PNX_HIP(ctx, hipMemsetAsync(ctx->d_band_probe.p, 0, (4 + LOOSE_MAX) * 4, ctx->s_pre));
    }
    {  // the flags of the groups and the bitmaps of the loose ones (BandLoose): zero when made, kept zero by the tail of every pass
        const size_t flag_bytes = ((size_t)ctx->n_groups + 1) * 4, bits_bytes = (size_t)LOOSE_MAX * ctx->n_blocks * BLOCK_WORDS * 4;
        // (by capacity, not by address: the allocator may hand a larger block out at the address of the one just freed)
        const size_t f0 = ctx->d_group_loose.cap, b0 = ctx->d_loose_bits.cap, e0 = ctx->d_entry_loose.cap;
        if ((rc = ensure(ctx, ctx->d_group_loose, flag_bytes)) || (rc = ensure(ctx, ctx->d_loose_bits, bits_bytes)) ||
            (rc = ensure(ctx, ctx->d_entry_loose, ((size_t)ctx->n_entries + 1) * 4)))
            return rc;
        if (ctx->d_entry_loose.cap != e0) PNX_HIP(ctx, hipMemsetAsync(ctx->d_entry_loose.p, 0, ctx->d_entry_loose.cap, ctx->s_pre));
        if (ctx->d_group_loose.cap != f0) PNX_HIP(ctx, hipMemsetAsync(ctx->d_group_loose.p, 0, ctx->d_group_loose.cap, ctx->s_pre));
        if (ctx->d_loose_bits.cap != b0 || ctx->loose_dirty) PNX_HIP(ctx, hipMemsetAsync(ctx->d_loose_bits.p, 0, ctx->d_loose_bits.cap, ctx->s_pre));
        ctx->loose_dirty = false;
    }
    // (a burst holds 1 .. 256 records; a list of single-step bursts is cut short by the directory: 1 entry per 4 records)
    if ((rc = ensure(ctx, ctx->d_spill, cap * 4)) || (rc = ensure(ctx, ctx->d_spill_dir, (cap / 4) * 8))) return rc;
    ctx->spill_cap = (uint32_t)cap;
    const bool fresh = ctx->d_spill_set.cap < slots * 8 || ctx->spill_slots != slots;
    if ((rc = ensure(ctx, ctx->d_spill_set, slots * 8))) return rc;
    ctx->spill_gen += 1;
    if (fresh || ctx->spill_gen > 255u) {  // a new table, or the generations have come round: every slot is empty again
        PNX_HIP(ctx, hipMemsetAsync(ctx->d_spill_set.p, 0, slots * 8, ctx->s_pre));
        ctx->spill_gen = 1;
    }
    ctx->spill_slots = slots;
    return PNX_OK;
}

// The entries of a one-shot pass: the visiting order, every path cut where the upload's summaries of its chunks say it turns
// round or jumps back (pass_pipeline.hip: path_cuts_from_chunks) -- the pieces of a path are entries of their own under the
// path's group.  Made once per (graph, order); without cuts it is the order itself with the paths' offsets written out, which
// saves the index kernel a dependent load.
int ensure_band_entries(pnx_ctx *ctx) {
    if (ctx->entries_valid) return PNX_OK;
    const uint32_t no = ctx->n_ordered;
    const bool cuts = ctx->h_cut_off.size() == (size_t)ctx->n_paths + 1 && !ctx->h_cuts.empty();
    std::vector<uint64_t> st, ln;
    std::vector<uint32_t> &gr = ctx->h_ent_group;
    gr.clear();
    st.reserve(no);
    ln.reserve(no);
    gr.reserve(no);
    auto push = [&](uint64_t a, uint64_t len, uint32_t g) { st.push_back(a), ln.push_back(len), gr.push_back(g); };
    for (uint32_t k = 0; k < no; ++k) {
        const uint32_t p = ctx->h_ord_path[k], g = ctx->h_ord_group[k];
        uint64_t a = ctx->h_path_off[p];
        uint64_t z = ctx->h_path_off[p + 1];
        if (ctx->h_sorted_at.size() == ctx->n_paths && ctx->h_sorted_at[p]) {  // a path that follows the ids nowhere: its sorted copy (upload_scan.hip)
            z = ctx->h_sorted_at[p] + (z - a);
            a = ctx->h_sorted_at[p];
        } else if (cuts) {
            for (uint32_t c = ctx->h_cut_off[p]; c < ctx->h_cut_off[p + 1]; ++c) {
                const uint64_t cut = ctx->h_cuts[c];
                if (cut > a && cut < z) {
                    push(a, cut - a, g);
                    a = cut;
                }
            }
        }
        push(a, z - a, g);
    }
    if (st.size() >= 0xFFFFFFFEull) return ctx->fail(PNX_ELIMIT, "too many pieces of paths in the visiting order");
    const size_t n = st.size();
    int rc;
    if ((rc = ensure(ctx, ctx->d_ent_start, (n + 1) * 8)) || (rc = ensure(ctx, ctx->d_ent_len, (n + 1) * 8)) || (rc = ensure(ctx, ctx->d_ent_group, (n + 1) * 4)))
        return rc;
    if (n) {
        PNX_HIP(ctx, hipMemcpyAsync(ctx->d_ent_start.p, st.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
        PNX_HIP(ctx, hipMemcpyAsync(ctx->d_ent_len.p, ln.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
        PNX_HIP(ctx, hipMemcpyAsync(ctx->d_ent_group.p, gr.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
        PNX_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (the host vectors go; other streams read the arrays)
    }
    ctx->n_entries = (uint32_t)n;
    ctx->entries_valid = true;
    return PNX_OK;
}

// Most (entry, band) cells empty -- fewer than 512 steps per cell on average: thousands of paths that each touch a few bands --:
// the workgroups walk per-band lists of the entries that have steps there (k_band_compact, k_band_cover<SPARSE>)
static bool band_sparse(const pnx_ctx *ctx, uint32_t n_bands) {
    if (const char *e = getenv("PNX_BAND_SPARSE")) {  // measurement: 0 / 1 forces one
        if ((e[0] == '0' || e[0] == '1') && e[1] == 0) return e[0] == '1';
    }
    return ctx->n_entries >= 64 && ctx->n_steps / ((uint64_t)n_bands * std::max<uint32_t>(ctx->n_entries, 1u)) < 512;
}

static BandLoose band_loose(const pnx_ctx *ctx) {
    return BandLoose{(uint32_t *)ctx->d_band_probe.p, (uint32_t *)ctx->d_group_loose.p, (uint32_t *)ctx->d_entry_loose.p, (uint32_t *)ctx->d_loose_bits.p};
}

template <int NPL>
static void launch_band_cover_t(pnx_ctx *ctx, bool write_m, uint32_t n_bands, const BandSplits &sp, bool sparse) {
    Ticket *tk = ctx->cur;
    const RowHist hs{tk->hist_fused ? (unsigned long long *)tk->d_hist_rep : nullptr,
                     ctx->weighted ? (const uint32_t *)ctx->d_weights.p : (const uint32_t *)nullptr, ctx->n_groups};
    const size_t lds_hist = tk->hist_fused ? ((size_t)ctx->n_groups + 1) * sizeof(unsigned long long) : 0;
    const BandSpill sl{(uint32_t *)ctx->d_spill.p, (unsigned long long *)ctx->d_spill_dir.p, ctx->spill_cap, ctx->spill_cap / 4};
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(n_bands * sp.n), dim3(BAND_CW * 64), lds_hist, ctx->s_main, (const uint32_t *)ctx->d_items.p,
                           (const unsigned long long *)tk->d_tile_idx_own.p, (const uint32_t *)ctx->d_ent_group.p, ctx->n_entries,
                           ctx->have_exclude ? (const uint8_t *)ctx->d_exclude.p : (const uint8_t *)nullptr, ctx->n_items, ctx->n_blocks,
                           (uint32_t *)ctx->d_M.p, (uint64_t)ctx->n_blocks * BLOCK_WORDS, (uint32_t *)tk->d_countable.p, hs, tk->d_flags,
                           n_bands, sp, sl, (const uint32_t *)ctx->d_band_probe.p, (const uint32_t *)tk->d_band_clist.p,
                           (const uint32_t *)tk->d_band_ccnt.p, (const uint32_t *)ctx->d_entry_loose.p);
    };
    // 4 waves per band, 2 loads in flight per lane: measured best on 10 M items x 256 and x 1024 paths (0.64 / 2.45 ms; 4 in flight
    // 0.67 / 2.56, 8 in flight 0.72; 8 waves per band 0.70, 2 waves 0.71) -- with every workgroup resident at once the chip holds
    // ~19 waves per CU whatever the register count, and a deeper pipeline only adds loads beyond the ends of the segments
    if (sparse) {
        if (sp.n > 1) {
            if (write_m) go(k_band_cover<NPL, BAND_CW, true, 2, true, true>);
            else go(k_band_cover<NPL, BAND_CW, false, 2, true, true>);
        } else {
            if (write_m) go(k_band_cover<NPL, BAND_CW, true, 2, false, true>);
            else go(k_band_cover<NPL, BAND_CW, false, 2, false, true>);
        }
    } else if (sp.n > 1) {
        if (write_m) go(k_band_cover<NPL, BAND_CW, true, 2, true, false>);
        else go(k_band_cover<NPL, BAND_CW, false, 2, true, false>);
    } else {
        if (write_m) go(k_band_cover<NPL, BAND_CW, true, 2, false, false>);
        else go(k_band_cover<NPL, BAND_CW, false, 2, false, false>);
    }
}

// phases 1 + 2 of a one-shot pass over the steps (its tail: launch_band_tail)
int launch_band_phases(pnx_ctx *ctx, bool write_m) {
    constexpr uint32_t BT = BAND_CW;
    Ticket *tk = ctx->cur;
    int rc;
    const uint32_t n_bands = (ctx->n_blocks + BT - 1) / BT;
    if ((rc = ensure_band_entries(ctx))) return rc;
    const uint64_t cells = (uint64_t)(n_bands + 1) * ctx->n_entries;
    if ((rc = ensure(ctx, tk->d_tile_idx_own, cells * 8)) || (rc = ensure(ctx, tk->d_group_first, ((size_t)ctx->n_groups + 1) * 4)) ||
        (rc = ensure_spill(ctx)))
        return rc;
    // the visiting order cut into splits of about as many entries each, at group boundaries
    BandSplits sp{};
    sp.n = ctx->band_splits ? ctx->band_splits : 1u;
    {
        const std::vector<uint32_t> &g = ctx->h_ent_group;
        const uint32_t no = ctx->n_entries;
        sp.k[0] = 0;
        for (uint32_t s = 1; s < sp.n; ++s) {
            uint32_t k = (uint32_t)((uint64_t)no * s / sp.n);
            while (k < no && k > 0 && g[k] == g[k - 1]) ++k;  // the first entry of the next group
            sp.k[s] = k < sp.k[s - 1] ? sp.k[s - 1] : k;
        }
        sp.k[sp.n] = no;
    }
    const bool sparse = band_sparse(ctx, n_bands);
    if (sparse) {
        if ((rc = ensure(ctx, tk->d_band_clist, (size_t)n_bands * ctx->n_entries * 4)) || (rc = ensure(ctx, tk->d_band_ccnt, (size_t)n_bands * sp.n * 4)))
            return rc;
        // groups without a step on a band are never met there: their part of the presence matrix is zero beforehand
        if (write_m) PNX_HIP(ctx, hipMemsetAsync(ctx->d_M.p, 0, (size_t)ctx->n_groups * ctx->n_blocks * BLOCK_WORDS * 4, ctx->s_pre));
    }
    const bool phased = ctx->s_pre != ctx->s_main;
    const uint64_t n_zero16 = sp.n > 1 ? ((uint64_t)ctx->n_items + 1 + 3) / 4 : 0;  // (the buffer is a multiple of 256 bytes)
    unsigned long long *dbg_time = nullptr;
    static const bool index_timing = getenv("PNX_BAND_INDEX_TIMING") != nullptr;
    static DevBuf d_dbg_time;  // (measurement only: one buffer per process, never freed)
    if (index_timing && ensure(ctx, d_dbg_time, 16 * 8) == PNX_OK) {
        dbg_time = (unsigned long long *)d_dbg_time.p;
        (void)hipMemsetAsync(dbg_time, 0, 16 * 8, ctx->s_pre);
    }
    prof_begin(ctx, PNX_K_INDEX, ctx->s_pre);
    hipLaunchKernelGGL(k_band_index, dim3((unsigned)((std::max<uint64_t>(cells, ctx->n_entries) + 255) / 256)), dim3(256), 0, ctx->s_pre,
                       (const uint32_t *)ctx->d_items.p, (const uint64_t *)ctx->d_ent_start.p, (const uint64_t *)ctx->d_ent_len.p,
                       (const uint32_t *)ctx->d_ent_group.p, ctx->n_entries, ctx->n_groups, n_bands, BT * BLOCK_ITEMS,
                       (unsigned long long *)tk->d_tile_idx_own.p, (uint32_t *)tk->d_group_first.p, (uint4 *)tk->d_block.p,
                       (uint32_t)(tk->block_bytes / 16), (uint4 *)tk->d_countable.p, n_zero16, (uint32_t *)ctx->d_band_probe.p,
                       (uint32_t *)ctx->d_group_loose.p, (uint32_t *)ctx->d_entry_loose.p, dbg_time);
    if (dbg_time) {  // (measurement only: waits for the kernel)
        unsigned long long t[16] = {};
        (void)hipStreamSynchronize(ctx->s_pre);
        (void)hipMemcpy(t, dbg_time, sizeof t, hipMemcpyDeviceToHost);
        fprintf(stderr, "[panacus_amd] k_band_index, one wave, us behind its start: entries %.2f, ends + sector %.2f, verdict %.2f, (branch) %.2f, search %.2f (%llu probes of the wave), statistics %.2f, filter + store %.2f\n",
                (t[1] - t[0]) / 100.0, (t[2] - t[0]) / 100.0, (t[3] - t[0]) / 100.0, (t[3] - t[0]) / 100.0, (t[4] - t[0]) / 100.0, t[8], (t[5] - t[0]) / 100.0, (t[6] - t[0]) / 100.0);
    }
    if (sparse)
        hipLaunchKernelGGL(k_band_compact, dim3((n_bands * sp.n + 3) / 4), dim3(256), 0, ctx->s_pre, (const unsigned long long *)tk->d_tile_idx_own.p,
                           n_bands, ctx->n_entries, sp, (uint32_t *)tk->d_band_clist.p, (uint32_t *)tk->d_band_ccnt.p);
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    // (recorded on one stream as well where the closed forms' tables are derived by the two-kernel route: that derivation starts
    // behind it -- kernels_closed_form.hip.  Otherwise nothing waits for it, and an event between the two kernels of one stream
    // costs the second one 6-15 us: index end -> coverage start 8-17 us with it, as seen in the kernel trace)
    tk->pre_recorded = phased || !quorum_route_fused(ctx->n_groups);
    if (tk->pre_recorded) PNX_HIP(ctx, hipEventRecord(tk->ev_pre, ctx->s_pre));
    if (phased) PNX_HIP(ctx, hipStreamWaitEvent(ctx->s_main, tk->ev_pre, 0));
    uint32_t bits = 1;  // planes needed to count up to n_groups inclusive
    while (bits < 32 && (ctx->n_groups >> bits) != 0) ++bits;
    prof_begin(ctx, PNX_K_COVER, ctx->s_main);
    if (bits <= 12) launch_band_cover_t<12>(ctx, write_m, n_bands, sp, sparse);
    else if (bits <= 24) launch_band_cover_t<24>(ctx, write_m, n_bands, sp, sparse);
    else {
        prof_end(ctx);
        return ctx->fail(PNX_ELIMIT, "more than 2^24-1 groups are not supported (got %u)", ctx->n_groups);
    }
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

// phase 3 of a one-shot pass: the spilled steps are added, then (a pass that added its histogram itself) the replicas are
// summed and [flags | hist] handed to the host -- one kernel where the rows route has k_hist_publish
int launch_band_tail(pnx_ctx *ctx, Ticket *tk, bool write_m) {
    constexpr uint32_t BT = BAND_CW;
    const uint32_t n_bands = (ctx->n_blocks + BT - 1) / BT;
    // (multi-GPU: the all-reduce of the block follows this kernel, the copy to the host follows that)
    const bool to_host = tk->hist_fused && tk->h_block_mapped && !(ctx->comm && ctx->comm_reduce_hist);
    tk->host_written = to_host;
    BandTail a{};
    a.items = (const uint32_t *)ctx->d_items.p;
    a.bidx = (const unsigned long long *)tk->d_tile_idx_own.p;
    a.group_first = (const uint32_t *)tk->d_group_first.p;
    a.exclude = ctx->have_exclude ? (const uint8_t *)ctx->d_exclude.p : nullptr;
    a.n_edges = n_bands + 1;
    a.n_items = ctx->n_items;
    a.sl = BandSpill{(uint32_t *)ctx->d_spill.p, (unsigned long long *)ctx->d_spill_dir.p, ctx->spill_cap, ctx->spill_cap / 4};
    a.scratch = tk->d_band_scratch;
    a.probe_stats = (uint32_t *)ctx->d_band_probe.p;
    a.hset = (unsigned long long *)ctx->d_spill_set.p;
    a.hmask = (uint32_t)(ctx->spill_slots - 1);
    a.gen = ctx->spill_gen;
    a.countable = (uint32_t *)tk->d_countable.p;
    a.M = write_m ? (uint32_t *)ctx->d_M.p : nullptr;
    a.row_words = (uint64_t)ctx->n_blocks * BLOCK_WORDS;
    a.hs = RowHist{tk->hist_fused ? (unsigned long long *)tk->d_hist_rep : nullptr,
                   ctx->weighted ? (const uint32_t *)ctx->d_weights.p : (const uint32_t *)nullptr, ctx->n_groups};
    a.hist = (unsigned long long *)tk->d_hist;
    a.flags = tk->d_flags;
    a.host_block = to_host ? (uint32_t *)tk->h_block_mapped : nullptr;
    // the scans may read half of what the pass itself reads (a burst scans the segments of EVERY path of its group: groups of two
    // haplotypes double the volume) -- and 4 M steps, microseconds, whatever the size of the graph
    a.scan_budget = (uint32_t)std::min<uint64_t>(ctx->n_steps / 2048 + 4096, 0xFFFFFFF0ull);
    a.lo = band_loose(ctx);
    a.ent_start = (const uint64_t *)ctx->d_ent_start.p;
    a.ent_len = (const uint64_t *)ctx->d_ent_len.p;
    a.n_tiles = ctx->n_blocks;
    a.n_groups = ctx->n_groups;
    a.n_ordered = ctx->n_entries;
    a.loose_budget = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ctx->n_steps / 8, 4ull << 20) >> 10, 0xFFFFFFF0ull);
    const size_t lds_hist = tk->hist_fused ? ((size_t)ctx->n_groups + 1) * sizeof(unsigned long long) : 0;
    // as many workgroups as the chip holds at once (the bursts are dealt round robin: workgroups that have to wait for a free
    // slot would start their share when the others have finished theirs -- 1024 workgroups on a chip that holds 768 of them, 3
    // per CU at 106 SGPRs, took two rounds where 768 take one), never fewer than the marking needs
    static thread_local int per_cu = 0;
    static thread_local size_t per_cu_lds = ~(size_t)0;
    if (per_cu_lds != lds_hist) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(k_band_tail), (int)(BAND_TAIL_WAVES * 64), lds_hist) != hipSuccess || n < 1) n = 2;
        per_cu = n;
        per_cu_lds = lds_hist;
    }
    uint32_t grid = (uint32_t)per_cu * (uint32_t)ctx->prop.multiProcessorCount;
    static const char *grid_env = getenv("PNX_BAND_TAIL_GRID");  // (measurement)
    if (grid_env) grid = (uint32_t)atoi(grid_env);
    grid = std::min<uint32_t>(std::max<uint32_t>(grid, LOOSE_WORKGROUPS), BAND_TAIL_GRID);
    prof_begin(ctx, PNX_K_HIST, ctx->s_post);
    hipLaunchKernelGGL(k_band_tail, dim3(grid), dim3(BAND_TAIL_WAVES * 64), lds_hist, ctx->s_post, a);
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

}  // namespace pnx

namespace pnx {
// pnx_preload: the first launch of a kernel loads the code object of its translation unit (tens of ms) and builds the
// kernel's function object; asking for a kernel's attributes does the same, without a launch -- and can be done by a host
// thread that has nothing else to do while the GFA text travels to HBM
void preload_band(unsigned what) {
    hipFuncAttributes a;
    auto touch = [&a](const void *k) { (void)hipFuncGetAttributes(&a, k); };
    if (what & PNX_PRELOAD_PASS) {
        touch((const void *)k_band_index);
        touch((const void *)k_band_cover<12, BAND_CW, false, 2, false, false>);
        touch((const void *)k_band_cover<24, BAND_CW, false, 2, false, false>);
        touch((const void *)k_band_tail);
    }
}
}  // namespace pnx
