// tile_counters.hpp -- what the coverage kernels share: the bit-sliced per-item counters of one item tile held by
// one wave, and the tail of a tile (coverage vector + histogram bins).
//
// AbacusByTotal::coverage (src/graph_broker/abacus.rs:719-744) counts, per item, the GROUPS that visit it; a wave
// that owns a tile of 2048 items holds those counts bit-sliced: plane k of lane L holds bit k of the count of the 32
// items behind word L (pnx_context.hpp: item n -> word n % 64, bit (n % 2048) / 64).  A finished group arrives as
// one presence word per lane.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pnx_context.hpp"

namespace pnx {

// The histogram of the pass, added by the coverage kernel itself (construct_hist / construct_hist_bps,
// src/graph_broker/abacus.rs:746-787): a workgroup collects the counts of its tiles in LDS and adds its non-empty bins to
// ONE OF HIST_REPLICAS copies of the histogram in global memory -- every tile ends with a burst of up to G+1 atomics, and on a
// single copy they would queue up on the few memory channels that hold it (4883 tiles x 257 bins on 2 KB).  k_hist_publish
// (kernels_hist.hip) adds the copies up.  rep == nullptr: the kernel only writes the coverage vector (K2 reads it).
struct RowHist {
    unsigned long long *rep;  // HIST_REPLICAS x (n_groups + 1)
    const uint32_t *weights;  // node lengths (bp), or nullptr
    uint32_t n_groups;
};

// A group is not rippled through all planes by itself: eight groups are first compressed by a tree of carry-save
// adders (3 inputs -> sum + carry, two 3-input bit operations each) into the three low planes and ONE carry word of
// weight 8, which alone ripples through the planes above -- 4 vector instructions per group instead of 3 NPL.
template <int NPL>
struct TileCounters {
    uint32_t cnt[NPL];
    uint32_t t0 = 0, tA = 0, fA = 0;  // pending: a single group, a carry of weight 2, a carry of weight 4
    uint32_t nfl = 0;                 // groups folded so far (wave-uniform)

    __device__ __forceinline__ TileCounters() {
#pragma unroll
        for (int k = 0; k < NPL; ++k) cnt[k] = 0;
    }
    __device__ static __forceinline__ void csa(uint32_t &hi, uint32_t &lo, uint32_t a, uint32_t b, uint32_t c) {
        const uint32_t u = a ^ b;
        hi = (a & b) | (u & c);
        lo = u ^ c;
    }
    __device__ __forceinline__ void ripple(uint32_t carry, int from) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            if (k >= from) {
                const uint32_t t = cnt[k] & carry;
                cnt[k] ^= carry;
                carry = t;
            }
        }
    }
    // one more group: x = its presence word (excluded items already removed)
    __device__ __forceinline__ void add_group(uint32_t x) {
        const uint32_t step = nfl & 7u;
        ++nfl;
        if ((step & 1u) == 0u) {
            t0 = x;
        } else if (step == 1u || step == 5u) {
            csa(tA, cnt[0], cnt[0], t0, x);
        } else {
            uint32_t tB, fB;
            csa(tB, cnt[0], cnt[0], t0, x);
            if (step == 3u) {
                csa(fA, cnt[1], cnt[1], tA, tB);
            } else {
                uint32_t e;
                csa(fB, cnt[1], cnt[1], tA, tB);
                csa(e, cnt[2], cnt[2], fA, fB);
                ripple(e, 3);
            }
        }
    }
    // fold what is still pending after the last group
    __device__ __forceinline__ void settle() {
        const uint32_t r = nfl & 7u;
        if (r & 1u) ripple(t0, 0);
        if (r & 2u) ripple(tA, 1);
        if (r & 4u) ripple(fA, 2);
    }
};

// The tail of a tile: the settled counters `cnt` of tile `tile` are unpacked into the coverage vector (the bit
// positions [b_lo, b_lo + n_bits) of every word: the waves that share a tile split them) and, with hs.rep, added to the
// workgroup's histogram bins in LDS (sh_hist: n_groups + 1 u64 bins, read as 4-byte bins for node counts; zeroed and
// flushed by the caller).  `own` = mask of this wave's bit positions.
// `word`: the presence word this lane holds (its items are word + 64 b); it is the lane itself except in k_rows_cover_q, whose
// lanes end up with the words in another order.
template <int NPL>
__device__ __forceinline__ void tile_tail(const uint32_t (&cnt)[NPL], uint32_t tile, uint32_t lane, uint32_t n_items,
                                          uint32_t *__restrict__ countable, const RowHist &hs, unsigned long long *sh_hist,
                                          uint32_t b_lo, uint32_t n_bits, uint32_t own, uint32_t word = 0xFFFFFFFFu) {
    if (word == 0xFFFFFFFFu) word = lane;
    uint32_t *sh32 = reinterpret_cast<uint32_t *>(sh_hist);  // node counts: 4-byte bins (a workgroup holds few tiles of 2048 items)
    const bool hist = hs.rep != nullptr, weighted = hs.weights != nullptr;
    // The bins 0, 1 and n_groups (uncovered, private and core items) hold most items of a pangenome; through LDS atomics
    // they would serialise up to 64 lanes on one address.  Node counts: the three bins are population counts of bit masks
    // over the planes -- 32 items of a lane at once --, only the other items go through LDS, one 4-byte add each.  Weighted
    // (bp): each lane keeps the three sums in registers.
    unsigned long long hot0 = 0, hot1 = 0, hotg = 0;
    uint32_t others = 0;
    if (hist && !weighted) {
        uint32_t any_hi = 0, any = 0, mg = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            any |= cnt[k];
            if (k >= 1) any_hi |= cnt[k];
            mg &= ((hs.n_groups >> k) & 1u) ? cnt[k] : ~cnt[k];
        }
        if (hs.n_groups <= 1 || (NPL < 32 && (hs.n_groups >> NPL) != 0)) mg = 0;  // (bin 1 / bin 0 take those items)
        uint32_t vm = 0;  // items 1 .. n_items of this word
        {
            const uint64_t first = (uint64_t)tile * BLOCK_ITEMS + word;  // item of bit 0; bit b: first + 64 b
            if (first <= n_items) {
                const uint64_t nb = (n_items - first) / 64 + 1;  // bits with an item <= n_items
                vm = nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u);
                if (first == 0) vm &= ~1u;  // item 0 is the sentinel
            }
        }
        vm &= own;
        const uint32_t m0 = ~any & vm, m1 = cnt[0] & ~any_hi & vm;
        mg &= vm & ~m1 & ~m0;
        hot0 = (uint32_t)__builtin_popcount(m0);
        hot1 = (uint32_t)__builtin_popcount(m1);
        hotg = (uint32_t)__builtin_popcount(mg);
        others = vm & ~(m0 | m1 | mg);
    }
    // unpack the bit-sliced counters: one coalesced 256-byte store per bit position (this wave's positions)
    for (uint32_t b = b_lo; b < b_lo + n_bits; ++b) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < NPL; ++k) v |= ((cnt[k] >> b) & 1u) << k;
        const uint64_t node = (uint64_t)tile * BLOCK_ITEMS + b * 64u + word;
        // countable[0] is the reference's reserved element (abacus.rs:549-551)
        if (node <= n_items) __builtin_nontemporal_store(node ? v : 0xFFFFFFFFu, countable + node);
        if (hist && !weighted) {
            if (((others >> b) & 1u) && v < hs.n_groups) atomicAdd(&sh32[v], 1u);  // abacus.rs:752: coverage beyond #groups is ignored
        } else if (hist && node >= 1 && node <= n_items) {
            const unsigned long long w = hs.weights[node];
            if (v == 0) hot0 += w;
            else if (v == 1) hot1 += w;
            else if (v == hs.n_groups) hotg += w;
            else if (v < hs.n_groups) atomicAdd(&sh_hist[v], w);  // abacus.rs:771
        }
    }
    if (hist) {
        for (int o = 32; o > 0; o >>= 1) {
            hot0 += __shfl_down(hot0, o);
            hot1 += __shfl_down(hot1, o);
            hotg += __shfl_down(hotg, o);
        }
        if (lane == 0) {
            if (weighted) {
                if (hot0) atomicAdd(&sh_hist[0], hot0);
                if (hot1) atomicAdd(&sh_hist[1], hot1);
                if (hotg) atomicAdd(&sh_hist[hs.n_groups], hotg);
            } else {
                if (hot0) atomicAdd(&sh32[0], (uint32_t)hot0);
                if (hot1) atomicAdd(&sh32[1], (uint32_t)hot1);
                if (hotg) atomicAdd(&sh32[hs.n_groups], (uint32_t)hotg);
            }
        }
    }
}

// after a barrier: the workgroup's bins go to one of the HIST_REPLICAS copies in global memory
__device__ __forceinline__ void hist_bins_flush(const RowHist &hs, const unsigned long long *sh_hist, uint32_t n_threads) {
    const uint32_t *sh32 = reinterpret_cast<const uint32_t *>(sh_hist);
    const bool weighted = hs.weights != nullptr;
    unsigned long long *dst = hs.rep + (size_t)(blockIdx.x % HIST_REPLICAS) * (hs.n_groups + 1);
    for (uint32_t b = threadIdx.x; b <= hs.n_groups; b += n_threads) {
        const unsigned long long x = weighted ? sh_hist[b] : (unsigned long long)sh32[b];
        if (x) atomicAdd(&dst[b], x);
    }
}

// exclusion word of a tile in presence layout (ActiveTable, src/util.rs:118-124)
__device__ __forceinline__ uint32_t tile_exclusion_word(const uint8_t *__restrict__ exclude, uint32_t tile, uint32_t lane, uint32_t n_items) {
    uint32_t excl = 0;
    if (exclude) {
        for (uint32_t b = 0; b < 32; ++b) {
            const uint64_t node = (uint64_t)tile * BLOCK_ITEMS + b * 64u + lane;
            if (node <= n_items && exclude[node]) excl |= 1u << b;
        }
    }
    return excl;
}

}  // namespace pnx
