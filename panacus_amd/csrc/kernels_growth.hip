// kernels_growth.hip -- ordered / permuted pangenome growth on the presence bit matrix.
//
// Replaces AbacusByGroup::calc_growth (src/graph_broker/abacus.rs:989-1032) for R group
// orders at once.  Per item i with ascending group ranks g_0 < g_1 < ... (under the order
// being evaluated) and total degree deg_i the reference adds w_i to res[j] for every
// j in [g_m, g_{m+1}) iff deg_i >= c and (m+1) >= ceil((g_m + 1) * q)   (abacus.rs:1001-1010),
// i.e. with x_j(i) = presence of i in the group of rank j:
//      cnt_j = cnt_{j-1} + x_j ;  ok_j = x_j ? (cnt_j >= Tq[j]) : ok_{j-1} ;
//      res[j] = sum_i w_i * [deg_i >= c] * ok_j(i).
// The quirk that the quorum bound uses the rank of the LAST containing group (not j) is
// exactly the "ok persists until the next set bit" rule above.
//
// Device formulation (integer set work, bit-parallel; no MFMA):
//   * one lane owns one u32 word = 32 items of a 2048-item block; a wave walks the ranks
//     of one order sequentially, reading one coalesced 256 B row slice per rank;
//   * q == 0 pairs:  ok_j = "seen so far", so only NEW bits (x & ~seen) change res: their
//     counts are deltas, prefix-summed by a finishing kernel;
//   * q > 0 pairs:   the slack cnt - Tq[j] is kept bit-sliced (a masked ripple per rank, or per
//     TWO ranks where the table rises at every other rank: q = 0.5) and only its sign is looked
//     at; the dense popcounts of a batch of 16 ranks are summed over the wave by a
//     reduce-scatter (v_permlane32_swap on the bits, v_permlane16_swap, DPP folds in the rows);
//   * per-workgroup LDS accumulators (u64 per rank) are flushed once with global atomics;
//     deltas of the q == 0 path are prefix-summed by a finishing kernel;
//   * bp counts use bit planes of the weights: sum_i w_i b_i = sum_p 2^p popc(b & W_p).
// Orders are independent, so R orders shard over workgroups (and over GPUs: permutation
// sharding, DESIGN.md "Multi-GPU").
#include <cstdlib>
#include <type_traits>

#include "pnx_context.hpp"

namespace pnx {

constexpr int GROW_WAVES = 4;
constexpr int GROW_PREFETCH = 16;  // row slices in flight per wave
constexpr int GROW_Q0_MAX = 4;     // q == 0 threshold pairs handled by one launch
constexpr int WPLANES_MAX = 32;
constexpr int GROW_EVQ = 128;  // bp: flip events queued per wave before they are applied
// bp: the weights of a block staged in LDS per wave (round 2) or read where an event needs one (round 6).  The kernel's bp
// variant is bound by how many workgroups a CU holds: 39 KB of LDS per workgroup with the staging (16 KB of it), 4 per CU; longer
// event queues made it slower (128 events: 18.7 ms, 256: 20.9, 512: 27.7 on cfg4) -- so the staging went: an event reads its
// weight from the block's 8 KB of the weight vector (L1 / L2 hits: the R orders of a block run side by side).
constexpr bool GROW_W_LDS = false;

// full-wave sum, result valid in lane 63 (gfx9 DPP: row_shr within rows of 16, then row_bcast)
__device__ static inline uint32_t wave_sum_to_lane63(uint32_t v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xe, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xc, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, true);  // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, true);  // row_bcast:31
    return v;
}

// ------------------------------------------------------------------------------------------
// threshold masks and weight planes in presence layout
// ------------------------------------------------------------------------------------------
// cmask[ci][blk][lane] bit b = countable[node] >= cvals[ci]
__global__ void k_cov_masks(const uint32_t *__restrict__ countable, uint32_t n_items, uint32_t n_blocks,
                            const uint32_t *__restrict__ cvals, uint32_t n_c, uint32_t *__restrict__ cmask) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t blk = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blk >= n_blocks) return;
    for (uint32_t ci = 0; ci < n_c; ++ci) {
        const uint32_t c = cvals[ci];
        uint32_t m = 0;
        for (uint32_t b = 0; b < 32; ++b) {
            uint64_t node = (uint64_t)blk * BLOCK_ITEMS + b * 64u + lane;
            if (node >= 1 && node <= n_items && countable[node] >= c) m |= 1u << b;
        }
        cmask[((uint64_t)ci * n_blocks + blk) * BLOCK_WORDS + lane] = m;
    }
}

__global__ void k_max_u32(const uint32_t *__restrict__ a, uint64_t first, uint64_t n, uint32_t *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + first;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t m = 0;
    for (; i < n; i += stride) m = a[i] > m ? a[i] : m;
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = __shfl_down(m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// wplanes[p][blk][lane] bit b = bit p of weights[node]
__global__ void k_weight_planes(const uint32_t *__restrict__ weights, uint32_t n_items, uint32_t n_blocks,
                                uint32_t n_planes, uint32_t *__restrict__ wplanes) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t blk = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blk >= n_blocks) return;
    uint32_t w[32];
#pragma unroll
    for (uint32_t b = 0; b < 32; ++b) {
        uint64_t node = (uint64_t)blk * BLOCK_ITEMS + b * 64u + lane;
        w[b] = (node >= 1 && node <= n_items) ? weights[node] : 0u;
    }
    for (uint32_t p = 0; p < n_planes; ++p) {
        uint32_t m = 0;
#pragma unroll
        for (uint32_t b = 0; b < 32; ++b) m |= ((w[b] >> p) & 1u) << b;
        wplanes[((uint64_t)p * n_blocks + blk) * BLOCK_WORDS + lane] = m;
    }
}

// weighted popcount through bit planes held in LDS (per wave: [plane][lane])
__device__ static inline unsigned long long weighted_popc(uint32_t bits, const uint32_t *wp_lds, uint32_t n_planes,
                                                          uint32_t lane) {
    unsigned long long s = 0;
    for (uint32_t p = 0; p < n_planes; ++p)
        s += (unsigned long long)__popc(bits & wp_lds[p * 64 + lane]) << p;
    return s;
}

// out rows flagged as "delta" become running sums over the ranks (wrapping u64 adds: the bp
// deltas of the quorum pairs may be negative).  One wave per row: 64 ranks per step, an inclusive
// scan over the lanes by shuffles, the carry in a scalar.
__global__ __launch_bounds__(64) void k_growth_prefix(unsigned long long *out, uint32_t G, uint32_t T,
                                                      const uint32_t *__restrict__ is_delta) {
    const uint32_t row = blockIdx.x;  // r * T + t
    if (!is_delta[row % T]) return;
    const uint32_t lane = threadIdx.x;
    unsigned long long *o = out + (uint64_t)row * G;
    unsigned long long carry = 0;
    for (uint32_t j0 = 0; j0 < G; j0 += 64) {
        const uint32_t j = j0 + lane;
        unsigned long long v = j < G ? o[j] : 0ull;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long u = __shfl_up(v, d);
            if (lane >= (uint32_t)d) v += u;
        }
        v += carry;
        if (j < G) o[j] = v;
        carry = __shfl(v, 63);
    }
}

// ------------------------------------------------------------------------------------------
// q > 0 pair (one per launch): bit-sliced running counts, dense popcount per rank
// ------------------------------------------------------------------------------------------
template <int NPL, bool WEIGHTED>
__global__ __launch_bounds__(GROW_WAVES * 64) void k_growth_quorum(
    const uint32_t *__restrict__ M, uint64_t row_words, uint32_t n_blocks, uint32_t G,
    const uint32_t *__restrict__ perms, uint32_t n_chunks, uint32_t blocks_per_chunk,
    const uint32_t *__restrict__ cmask_t /* mask of this pair or nullptr */, const uint32_t *__restrict__ qtab_t,
    uint32_t t_slot, uint32_t T, const uint32_t *__restrict__ wplanes, uint32_t n_planes,
    unsigned long long *out) {
    extern __shared__ unsigned long long smem[];
    unsigned long long *acc = smem;  // [G]
    uint32_t *wp_all = reinterpret_cast<uint32_t *>(acc + G);

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t r = blockIdx.x / n_chunks, chunk = blockIdx.x % n_chunks;
    const uint32_t *perm = perms + (uint64_t)r * G;
    uint32_t *wp = wp_all + (size_t)wave * WPLANES_MAX * 64;

    for (uint32_t i = threadIdx.x; i < G; i += blockDim.x) acc[i] = 0;
    __syncthreads();

    const uint32_t b_end = min(n_blocks, (chunk + 1) * blocks_per_chunk);
    for (uint32_t blk = chunk * blocks_per_chunk + wave; blk < b_end; blk += GROW_WAVES) {
        const uint32_t mask = cmask_t ? cmask_t[(uint64_t)blk * BLOCK_WORDS + lane] : 0xFFFFFFFFu;
        if (WEIGHTED) {
            for (uint32_t p = 0; p < n_planes; ++p)
                wp[p * 64 + lane] = wplanes[((uint64_t)p * n_blocks + blk) * BLOCK_WORDS + lane];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        const uint32_t *col = M + (uint64_t)blk * BLOCK_WORDS + lane;
        uint32_t cnt[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) cnt[k] = 0;
        uint32_t ok = 0;
        for (uint32_t jb = 0; jb < G; jb += GROW_PREFETCH) {
            uint32_t x[GROW_PREFETCH];
#pragma unroll
            for (int u = 0; u < GROW_PREFETCH; ++u) {
                x[u] = 0;
                if (jb + u < G) x[u] = col[(uint64_t)perm[jb + u] * row_words];
            }
#pragma unroll
            for (int u = 0; u < GROW_PREFETCH; ++u) {
                if (jb + u < G) {  // wave-uniform
                    const uint32_t thr = qtab_t[jb + u];  // wave-uniform (scalar)
                    // cnt += x  (ripple-carry over the planes)
                    uint32_t carry = x[u];
#pragma unroll
                    for (int k = 0; k < NPL; ++k) {
                        const uint32_t tmp = cnt[k] & carry;
                        cnt[k] ^= carry;
                        carry = tmp;
                    }
                    // ge = (cnt >= thr), most significant plane first; thr is uniform
                    uint32_t gt = 0, eq = 0xFFFFFFFFu;
#pragma unroll
                    for (int k = NPL - 1; k >= 0; --k) {
                        const uint32_t tb = 0u - ((thr >> k) & 1u);  // all-ones if bit k of thr is set
                        gt |= eq & cnt[k] & ~tb;
                        eq &= ~(cnt[k] ^ tb);
                    }
                    const uint32_t ge = gt | eq;
                    ok = (ok & ~x[u]) | (ge & x[u]);
                    const uint32_t bits = ok & mask;
                    if (WEIGHTED) {
                        unsigned long long s = weighted_popc(bits, wp, n_planes, lane);
                        uint32_t lo = wave_sum_to_lane63((uint32_t)(s & 0xFFFFFFu));         // 24 + 6 bits
                        uint32_t mi = wave_sum_to_lane63((uint32_t)((s >> 24) & 0xFFFFFFu));
                        if (lane == 63) atomicAdd(&acc[jb + u], (unsigned long long)lo + ((unsigned long long)mi << 24));
                    } else {
                        const uint32_t tot = wave_sum_to_lane63((uint32_t)__popc(bits));
                        if (lane == 63 && tot) atomicAdd(&acc[jb + u], (unsigned long long)tot);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < G; j += blockDim.x) {
        const unsigned long long v = acc[j];
        if (v) atomicAdd(&out[((uint64_t)r * T + t_slot) * G + j], v);
    }
}

// ------------------------------------------------------------------------------------------
// fused growth kernel: N0 pairs with q == 0 and NQ pairs with q > 0 share ONE read of the
// presence rows of an order.
//   * q > 0 pairs use the slack form s_j = cnt_j - Tq[j], bit-sliced.  Tq rises by dT in {0,1} per rank (q <= 1), so per
//     rank s += x (dT = 0) or s += x - 1 (dT = 1): ONE ripple pass under mask m = x ^ dmask with the plane complemented by
//     dmask (dmask = 0 / ~0 is wave-uniform and comes from a host table).  The pass runs over the SIX low planes only:
//     s = 32 H + L, L in [0, 63] is rippled per rank (or per two ranks: ALT), H (two's complement) once per batch of 16
//     ranks when L is brought back into [16, 47]; "cnt >= Tq" is (H >= 0) | (H == -1 & L >= 32) with the two H masks kept
//     between those steps (round 3: five planes, H every eight ranks).
//   * everything on the per-rank path is branch-free VALU: the first version of this kernel
//     was bound by the CU's single scalar unit (27 SALU instructions per rank for uniform
//     branches, exec-mask juggling and 64-bit address arithmetic); row offsets now come
//     pre-multiplied from the host as 32-bit byte offsets (saddr + voffset addressing).
//   * per-rank popcounts are summed over the wave once per batch of 16 ranks by a reduce-scatter (see the batch loop):
//     round 2 took six DPP steps on each of 24 registers and a round trip through LDS (18.4 ms per 128 orders of cfg4, 8.5
//     with the sums thrown away), round 3 v_permlane32_swap / v_permlane16_swap + four DPP steps per row register (15.8 ms),
//     round 5 starts on the bits and scatters inside the rows as well (11.8 ms with the two-rank step; DESIGN.md K4: the
//     kernel's time is its vector instructions, 481 -> 332 per batch).
// Workgroups of one block chunk carry consecutive blockIdx for all R orders.  (Giving all R orders of a
// chunk to ONE XCD -- chunk % 8, so that its L2 serves a row to R orders for one fetch -- was measured:
// 18.8 against 18.2 ms on cfg4; the kernel is bound by VALU issue, not by the rows.)
// ------------------------------------------------------------------------------------------
struct GrowthTabs {
    int32_t q0_midx[GROW_Q0_MAX];   // mask index per q == 0 pair, -1 = none
    uint32_t q0_slot[GROW_Q0_MAX];  // output slot t
    int32_t qq_midx[2];
    uint32_t qq_slot[2];
};

// WMODE: 0 = items count 1, 1 = weights below 2^16 (staged as u16: 4 KB of LDS per wave),
//        2 = any u32 weights (8 KB per wave)
// ALT: every pair of ranks of every quorum table of the launch is d = (1, 0)  (q = 0.5): the short pair step
template <int NPL1, int N0, int NQ, int WMODE, bool ALT>
__global__ __launch_bounds__(GROW_WAVES * 64, (WMODE == 0 && NPL1 <= 11) ? 6 : 1) void k_growth_fused(
    const uint32_t *__restrict__ M, uint32_t n_blocks, uint32_t G,
    const uint32_t *__restrict__ rowoff /* R x G byte offsets of the rows */, uint32_t R,
    uint32_t blocks_per_chunk, const uint32_t *__restrict__ cmask, GrowthTabs tabs,
    const uint32_t *__restrict__ dmask /* T x G: 0 or ~0 */, uint32_t T,
    const uint32_t *__restrict__ weights, uint32_t n_items, unsigned long long *out, uint32_t evq_n /* bp: events a wave queues */) {
    constexpr int NA = N0 + NQ;
    constexpr bool WEIGHTED = WMODE != 0;
    constexpr int B = GROW_PREFETCH;  // ranks per batch
    extern __shared__ unsigned long long smem[];
    unsigned long long *acc = smem;                                           // [NA][G]
    uint32_t *stage_all = reinterpret_cast<uint32_t *>(acc + (size_t)NA * G);  // [waves][NA][B/2]
    uint32_t *wp_all = stage_all + GROW_WAVES * NA * (B / 2);                  // [waves][32][64] item weights

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t r = blockIdx.x % R, chunk = blockIdx.x / R;
    const uint32_t *ro = rowoff + (uint64_t)r * G;
    uint32_t *stage = stage_all + wave * NA * (B / 2);
    constexpr uint32_t WSTAGE = GROW_W_LDS ? (WMODE == 1 ? 1024u : 2048u) : 0u;  // words of staged weights per wave
    uint32_t *wp = wp_all + (size_t)wave * WSTAGE;
    uint16_t *wp16 = reinterpret_cast<uint16_t *>(wp);
    constexpr int EVW = 1 + NA + NQ;  // event: (rank << 8 | lane), up mask per accumulator, down mask per quorum pair
    uint32_t *evq = wp_all + (size_t)GROW_WAVES * WSTAGE + (size_t)wave * evq_n * EVW;
    uint32_t qn = 0;  // events in the queue (wave-uniform)
    // the presence matrix as a buffer (below 4 GiB on this route): a row load is base + the row's byte offset (a scalar) +
    // the lane's offset within the row, no vector instruction for the address
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(M), 0, -1, 0x00020000);

    for (uint32_t i = threadIdx.x; i < (uint32_t)NA * G; i += blockDim.x) acc[i] = 0;
    __syncthreads();

    // where the lane's total of the in-row reduce-scatter below goes: the register that ends in this lane (walking the four
    // levels back: a lane's bit chooses the first or the second of a folded pair), the accumulator and the ranks that
    // register carries; where a register folded without a partner the lanes with the bit set hold a copy and stay out
    const bool rs_sel8 = (lane & 8u) != 0, rs_sel4 = (lane & 4u) != 0, rs_sel2 = (lane & 2u) != 0, rs_sel1 = (lane & 1u) != 0;
    bool rs_writer = !WEIGHTED;
    uint32_t rs_acc = 0;
    if (!WEIGHTED) {
        constexpr int H1 = (NA > 0 ? NA : 1) * (B / 4), H2 = H1 / 2;  // (NA = 0 is compiled, never launched)
        constexpr int nin[4] = {H2, (H2 + 1) / 2, ((H2 + 1) / 2 + 1) / 2, (((H2 + 1) / 2 + 1) / 2 + 1) / 2};
        static_assert((nin[3] + 1) / 2 == 1, "four levels bring the registers of a row down to one");
        uint32_t pos = 0;
#pragma unroll
        for (int L = 3; L >= 0; --L) {
            const uint32_t bit = (lane >> (3 - L)) & 1u;
            if (2 * pos + 1 < (uint32_t)nin[L]) {
                pos = 2 * pos + bit;
            } else {
                pos = 2 * pos;
                if (bit) rs_writer = false;
            }
        }
        // row rho: even rows hold the registers i < H2 of the packed counts, odd rows i + H2; register a * 4 + i carries, of
        // accumulator a, the pairs of ranks i (low half) and i + 4 (high half): the even rank in the rows 0, 1 (the lanes below
        // 32), the odd one in the rows 2, 3
        const uint32_t rho = lane >> 4;
        const uint32_t r = (uint32_t)H2 * (rho & 1u) + pos;
        rs_acc = (r >> 2) * G + 2u * (r & 3u) + (rho >> 1);
    }

    const uint32_t b_end = min(n_blocks, (chunk + 1) * blocks_per_chunk);
    for (uint32_t blk = chunk * blocks_per_chunk + wave; blk < b_end; blk += GROW_WAVES) {
        uint32_t mask[NA > 0 ? NA : 1];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            const int32_t mi = a < N0 ? tabs.q0_midx[a] : tabs.qq_midx[a - N0];
            mask[a] = 0xFFFFFFFFu;
            if (mi >= 0) mask[a] = cmask[((uint64_t)mi * n_blocks + blk) * BLOCK_WORDS + lane];
        }
        if (WEIGHTED && GROW_W_LDS) {
            // weights of this block in presence layout: item (bit b, lane) at wp[b * 64 + lane]
            for (uint32_t b = 0; b < 32; ++b) {
                const uint64_t node = (uint64_t)blk * BLOCK_ITEMS + b * 64u + lane;
                const uint32_t wv = (node >= 1 && node <= n_items) ? weights[node] : 0u;
                if (WMODE == 1) wp16[b * 64 + lane] = (uint16_t)wv; else wp[b * 64 + lane] = wv;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        // bp: every accumulator is kept in DELTA form -- an item only changes res[j] at a rank
        // where its bit of `val` flips (q = 0: once, when it is first seen; q > 0: where ok
        // flips), so the weights of the few flipping items are added (or, wrapping, subtracted)
        // with LDS atomics and a finishing kernel takes the running sums.
        uint32_t prevv[NQ > 0 ? NQ : 1];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) prevv[qi] = 0;
        // A lane whose masks flip at this rank queues ONE event (rank, lane, masks); the queue is
        // applied by all 64 lanes in parallel -- one event per lane -- when it fills up and at the
        // end of the block.  The rank loop itself stays free of per-lane loops: flips are rare
        // (about two events per item and order), so doing them where they occur would make every
        // rank pay for the slowest lane.
        auto apply_events = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t e = lane; e < qn; e += 64) {
                const uint32_t *ev = evq + (size_t)e * EVW;
                const uint32_t hdr = ev[0];
                const uint32_t j = hdr >> 8, el = hdr & 63u;
                uint32_t up[NA > 0 ? NA : 1], dn[NQ > 0 ? NQ : 1];
                uint32_t any = 0;
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    up[a] = ev[1 + a];
                    any |= up[a];
                }
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) {
                    dn[qi] = ev[1 + NA + qi];
                    any |= dn[qi];
                }
                unsigned long long d[NA > 0 ? NA : 1];
#pragma unroll
                for (int a = 0; a < NA; ++a) d[a] = 0;
                while (any) {
                    const uint32_t b = (uint32_t)__builtin_ctz(any);
                    any &= any - 1;
                    // (a set bit is an item of the graph: its id is within 1 .. n_items)
                    const unsigned long long w = !GROW_W_LDS ? weights[(uint64_t)blk * BLOCK_ITEMS + b * 64u + el]
                                                 : (WMODE == 1 ? (uint32_t)wp16[b * 64 + el] : wp[b * 64 + el]);
#pragma unroll
                    for (int a = 0; a < NA; ++a) d[a] += ((up[a] >> b) & 1u) ? w : 0ull;
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) d[N0 + qi] -= ((dn[qi] >> b) & 1u) ? w : 0ull;
                }
#pragma unroll
                for (int a = 0; a < NA; ++a)
                    if (d[a]) atomicAdd(&acc[(size_t)a * G + j], d[a]);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            qn = 0;
        };
        auto weighted_rank = [&](const uint32_t (&val)[NA > 0 ? NA : 1], uint32_t j) {
            uint32_t up[NA > 0 ? NA : 1], dn[NQ > 0 ? NQ : 1];
            uint32_t any = 0;
#pragma unroll
            for (int a = 0; a < N0; ++a) {
                up[a] = val[a];
                any |= up[a];
            }
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                const uint32_t cur = val[N0 + qi];
                up[N0 + qi] = cur & ~prevv[qi];
                dn[qi] = prevv[qi] & ~cur;
                prevv[qi] = cur;
                any |= up[N0 + qi] | dn[qi];
            }
            const unsigned long long bal = __ballot(any != 0);
            if (bal) {
                if (any) {
                    const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    uint32_t *ev = evq + (size_t)pos * EVW;
                    ev[0] = (j << 8) | lane;
#pragma unroll
                    for (int a = 0; a < NA; ++a) ev[1 + a] = up[a];
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) ev[1 + NA + qi] = dn[qi];
                }
                qn += (uint32_t)__builtin_popcountll(bal);
                if (qn > evq_n - 64u) apply_events();
            }
        };
        const uint32_t voff = (blk * BLOCK_WORDS + lane) * 4u;  // byte offset of this lane's word in a row
        uint32_t seen = 0;
        // The slack s = cnt - Tq(j) of every item, bit-sliced, in TWO parts: s = 32 H + L with L in [0, 63] (six planes) and
        // H (NPLH planes, two's complement).  A rank moves s by at most one, so only the planes of L are rippled per rank;
        // once per batch of 16 ranks L is brought back into [16, 47] by moving 32 into or out of H -- one ripple over H per
        // batch -- and between two such steps H does not change: s >= 0 <=> H >= 0, or H == -1 and L >= 32, with "H >= 0"
        // and "H == -1" kept as masks.  The pair's coverage mask is folded into those two, so that `ok` never leaves it.
        // (Round 3 / 4: five planes, H every eight ranks; with two ranks per ripple a plane of L costs one instruction
        // per rank and a step over H three and a half.)
        constexpr int NLO = 6, NPLH = NPL1 - 4;
        uint32_t lo[NQ > 0 ? NQ : 1][NLO], hi[NQ > 0 ? NQ : 1][NPLH], hpos[NQ > 0 ? NQ : 1], hm1[NQ > 0 ? NQ : 1];
        uint32_t ok[NQ > 0 ? NQ : 1];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {  // s = 0: H = -1, L = 32
            ok[qi] = 0;
#pragma unroll
            for (int k = 0; k < NLO - 1; ++k) lo[qi][k] = 0;
            lo[qi][NLO - 1] = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < NPLH; ++k) hi[qi][k] = 0xFFFFFFFFu;
            hpos[qi] = 0;
            hm1[qi] = mask[N0 + qi];
        }
        auto renorm = [&]() {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                const uint32_t l4 = lo[qi][NLO - 2], l5 = lo[qi][NLO - 1];
                const uint32_t dn = ~(l5 | l4);  // L < 16: L += 32, H -= 1
                uint32_t m = (l5 & l4) | dn;     // L >= 48: L -= 32, H += 1
                lo[qi][NLO - 1] = ~l4;
                uint32_t all = mask[N0 + qi];
#pragma unroll
                for (int k = 0; k < NPLH; ++k) {
                    const uint32_t t = (hi[qi][k] ^ dn) & m;  // carry where the bit was 1 (up), borrow where it was 0 (down)
                    hi[qi][k] ^= m;
                    m = t;
                    all &= hi[qi][k];
                }
                hpos[qi] = ~hi[qi][NPLH - 1] & mask[N0 + qi];
                hm1[qi] = all;
            }
        };

        // one rank: returns the NA per-lane contributions
        auto rank_step = [&](uint32_t xv, uint32_t j, uint32_t (&val)[NA > 0 ? NA : 1]) {
            if (N0 > 0) {
                const uint32_t nw = xv & ~seen;
                seen |= xv;
#pragma unroll
                for (int a = 0; a < N0; ++a) val[a] = nw & mask[a];
            }
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                const uint32_t dm = dmask[(uint64_t)tabs.qq_slot[qi] * G + j];  // wave-uniform
                uint32_t m = xv ^ dm;  // dT = 0: increment where x ; dT = 1: decrement where !x
#pragma unroll
                for (int k = 0; k < NLO; ++k) {
                    const uint32_t t = (lo[qi][k] ^ dm) & m;
                    lo[qi][k] ^= m;
                    m = t;
                }
                const uint32_t ge = hpos[qi] | (hm1[qi] & lo[qi][NLO - 1]);  // s >= 0  <=>  cnt >= Tq  (within the pair's mask)
                ok[qi] = (ok[qi] & ~xv) | (ge & xv);
                val[N0 + qi] = ok[qi];
            }
        };

        // TWO ranks a = j, b = j + 1 in one step where every pair of ranks of the quorum tables is d = (1, 0) -- q = 0.5
        // (round 5): the kernel's time is its vector instructions (30 per rank at cfg4, 20 of them the quorum pair's), and
        // two ranks share ONE pass over the planes of L.  The slack moves by (x_a - 1) + x_b: +1 where both bits are set,
        // -1 where neither is, nothing where one is -- the masked ripple of the single step with a per-item direction.
        // `ok` after rank a needs "s_a >= 0" only where x_a is set, and there s_a = s_prev + 1 - d_a = s_prev: the mask
        // "s >= 0" as it stands before the step.  (The same for any table -- the ripple starting at plane 0 or 1 by
        // |x_a + x_b - d_a - d_b|, "s == -1" beside "s >= 0" where d_a = 0 -- was written and counted: 494 instructions per
        // batch against 481 for single steps; such tables keep the single step.)
        auto pair_step = [&](uint32_t xa, uint32_t xb, uint32_t (&va)[NA > 0 ? NA : 1], uint32_t (&vb)[NA > 0 ? NA : 1]) {
            if (N0 > 0) {
                const uint32_t nwa = xa & ~seen;
                seen |= xa;
                const uint32_t nwb = xb & ~seen;
                seen |= xb;
#pragma unroll
                for (int a = 0; a < N0; ++a) {
                    va[a] = nwa & mask[a];
                    vb[a] = nwb & mask[a];
                }
            }
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                uint32_t *L = lo[qi];
                const uint32_t gp = hpos[qi] | (hm1[qi] & L[NLO - 1]);  // s_prev >= 0
                const uint32_t oka = (ok[qi] & ~xa) | (gp & xa);
                va[N0 + qi] = oka;
                const uint32_t h = xa & xb;
                uint32_t m = ~(xa ^ xb);
#pragma unroll
                for (int k = 0; k < NLO; ++k) {
                    const uint32_t t = ~(L[k] ^ h) & m;  // carry where both (the bit was 1), borrow where neither (it was 0)
                    L[k] ^= m;
                    m = t;
                }
                const uint32_t ge = hpos[qi] | (hm1[qi] & L[NLO - 1]);  // s_b >= 0
                ok[qi] = (oka & ~xb) | (ge & xb);
                vb[N0 + qi] = ok[qi];
            }
        };

        uint32_t jb = 0;
        for (; jb + B <= G; jb += B) {  // full batches: no guards on the per-rank path
            uint32_t x[B];
#pragma unroll
            for (int u = 0; u < B; ++u) x[u] = __builtin_amdgcn_raw_buffer_load_b32(mrs, voff, ro[jb + u], 0);
            if (!WEIGHTED) {
                // The per-rank counts are summed over the 64 lanes by a reduce-SCATTER that starts on the BITS: v_permlane32_swap
                // puts the upper half of rank u's mask beside the lower half of rank u + 1's, so the two population counts of
                // a lane (the second one adds the first: v_bcnt's free operand) are already folded over lane ^ 32 -- rank u in
                // lanes 0..31, rank u + 1 in lanes 32..63 -- for the three instructions that two counts and their packing
                // cost before.  Pairs p and p + 4 then share a register (16 bits each: a half-wave's count is at most 2048),
                // v_permlane16_swap folds register i beside register i + H2 over the 16-lane rows.
                uint32_t pk[NA > 0 ? NA : 1][B / 2];
                if (ALT) {
#pragma unroll
                    for (int u = 0; u < B; u += 2) {
                        uint32_t va[NA > 0 ? NA : 1], vb[NA > 0 ? NA : 1];
                        pair_step(x[u], x[u + 1], va, vb);
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(va[a], vb[a], false, false);
                            pk[a][u >> 1] = (uint32_t)__popc(sw[0]) + (uint32_t)__popc(sw[1]);
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < B; u += 2) {
                        uint32_t va[NA > 0 ? NA : 1], vb[NA > 0 ? NA : 1];
                        rank_step(x[u], jb + u, va);
                        rank_step(x[u + 1], jb + u + 1, vb);
#pragma unroll
                        for (int a = 0; a < NA; ++a) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(va[a], vb[a], false, false);
                            pk[a][u >> 1] = (uint32_t)__popc(sw[0]) + (uint32_t)__popc(sw[1]);
                        }
                    }
                }
                if (NQ > 0) renorm();
                {
                    constexpr int NA1 = NA > 0 ? NA : 1, H1 = NA1 * (B / 4), H2 = H1 / 2;  // (NA = 0 is compiled, never launched)
                    static_assert(B == 16, "the reduce-scatter below takes 8 pairs of ranks per accumulator");
                    uint32_t w1[H1], w2[H2];
#pragma unroll
                    for (int a = 0; a < NA; ++a)
#pragma unroll
                        for (int i = 0; i < B / 4; ++i) w1[a * (B / 4) + i] = pk[a][i] | (pk[a][i + B / 4] << 16);
#pragma unroll
                    for (int i = 0; i < H2; ++i) {  // even rows: register i of w1, odd rows: register i + H2, folded over lane ^ 16
                        const auto sw = __builtin_amdgcn_permlane16_swap(w1[i], w1[i + H2], false, false);
                        w2[i] = sw[0] + sw[1];
                    }
                    // Within the rows the reduce-scatter goes on (round 5): two registers fold into one over lane ^ 8 -- the lanes
                    // with bit 3 clear keep the first, the others the second: two selects and one addition with a DPP operand
                    // (row_ror:8) --, then over the mirrored half rows, the reversed quads, the neighbours; a register without a
                    // partner folds on its own.  2 NA registers take 17 instructions at NA = 3 instead of 24, and every total
                    // ends in a lane of its own (rs_acc: where it goes, worked out once per kernel), so that ONE pair of LDS
                    // additions serves the row instead of one per register.
                    constexpr int n0 = H2, n1 = (n0 + 1) / 2, n2 = (n1 + 1) / 2, n3 = (n2 + 1) / 2;
                    uint32_t a1[n1], a2[n2], a3[n3], a4;
#define PNX_RS_LEVEL(IN, NIN, OUT, CTRL, SEL)                                                                        \
    _Pragma("unroll") for (int i = 0; i < (NIN) / 2; ++i) {                                                          \
        const uint32_t x = (SEL) ? IN[2 * i + 1] : IN[2 * i], y = (SEL) ? IN[2 * i] : IN[2 * i + 1];                 \
        OUT[i] = x + (uint32_t)__builtin_amdgcn_update_dpp(0u, y, CTRL, 0xf, 0xf, true);                             \
    }                                                                                                                \
    if ((NIN) & 1) OUT[(NIN) / 2] = IN[(NIN) - 1] + (uint32_t)__builtin_amdgcn_update_dpp(0u, IN[(NIN) - 1], CTRL, 0xf, 0xf, true);
                    PNX_RS_LEVEL(w2, n0, a1, 0x128, rs_sel8)  // row_ror:8
                    PNX_RS_LEVEL(a1, n1, a2, 0x141, rs_sel4)  // row_half_mirror
                    PNX_RS_LEVEL(a2, n2, a3, 0x01B, rs_sel2)  // quad_perm:[3,2,1,0]
                    {
                        uint32_t *a4p = &a4;
                        PNX_RS_LEVEL(a3, n3, a4p, 0x0B1, rs_sel1)  // quad_perm:[1,0,3,2]
                    }
#undef PNX_RS_LEVEL
                    if (rs_writer) {  // the lane's register carries one rank of pair p (low half) and one of pair p + 4
                        unsigned long long *dst = &acc[rs_acc + jb];
                        atomicAdd(dst, (unsigned long long)(a4 & 0xFFFFu));  // a test for zero would cost more than the add
                        atomicAdd(dst + B / 2, (unsigned long long)(a4 >> 16));
                    }
                }
            } else {
                if (ALT) {
#pragma unroll
                    for (int u = 0; u < B; u += 2) {
                        uint32_t va[NA > 0 ? NA : 1], vb[NA > 0 ? NA : 1];
                        pair_step(x[u], x[u + 1], va, vb);
                        weighted_rank(va, jb + u);
                        weighted_rank(vb, jb + u + 1);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < B; ++u) {
                        uint32_t val[NA > 0 ? NA : 1];
                        rank_step(x[u], jb + u, val);
                        weighted_rank(val, jb + u);
                    }
                }
                if (NQ > 0) renorm();
            }
        }
        for (; jb < G; ++jb) {  // tail ranks (G not a multiple of the batch: fewer than 16, L stays within its planes)
            const uint32_t xv = __builtin_amdgcn_raw_buffer_load_b32(mrs, voff, ro[jb], 0);
            uint32_t val[NA > 0 ? NA : 1];
            rank_step(xv, jb, val);
            if (WEIGHTED) {
                weighted_rank(val, jb);
            } else {
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const uint32_t tot = wave_sum_to_lane63((uint32_t)__popc(val[a]));
                    if (lane == 63 && tot) atomicAdd(&acc[(size_t)a * G + jb], (unsigned long long)tot);
                }
            }
        }
        if (WEIGHTED && qn) apply_events();  // the next block brings its own weights
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t)NA * G; i += blockDim.x) {
        const unsigned long long v = acc[i];
        if (v) {
            const uint32_t a = i / G, j = i % G;
            const uint32_t t = a < (uint32_t)N0 ? tabs.q0_slot[a] : tabs.qq_slot[a - N0];
            atomicAdd(&out[((uint64_t)r * T + t) * G + j], v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// bit planes of the resident weights in presence layout (built once per upload)
int ensure_weight_planes(pnx_ctx *ctx, uint32_t *d_scratch) {
    if (ctx->wplanes_valid) return PNX_OK;
    const uint32_t NB = ctx->n_blocks;
    int rc;
    DevBuf tmp;
    uint32_t *d_max = d_scratch;
    if (!d_max) {
        if ((rc = ensure(ctx, tmp, sizeof(uint32_t)))) return rc;
        d_max = (uint32_t *)tmp.p;
    }
    hipError_t e = hipMemsetAsync(d_max, 0, sizeof(uint32_t), ctx->stream);
    uint32_t mx = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_max_u32, dim3(1024), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_weights.p,
                           (uint64_t)1, (uint64_t)ctx->n_items + 1, d_max);
        e = hipMemcpyAsync(&mx, d_max, sizeof mx, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    release(tmp);
    if (e != hipSuccess) return ctx->fail(PNX_EHIP, "weight planes: %s", hipGetErrorString(e));
    uint32_t planes = 0;
    while (planes < 32 && (mx >> planes) != 0) ++planes;
    if (planes == 0) planes = 1;
    ctx->n_wplanes = planes;
    if ((rc = ensure(ctx, ctx->d_wplanes, (size_t)planes * NB * BLOCK_WORDS * sizeof(uint32_t)))) return rc;
    if (NB)
        hipLaunchKernelGGL(k_weight_planes, dim3((NB + 3) / 4), dim3(256), 0, ctx->stream,
                           (const uint32_t *)ctx->d_weights.p, ctx->n_items, NB, planes, (uint32_t *)ctx->d_wplanes.p);
    ctx->wplanes_valid = true;
    return PNX_OK;
}

int launch_growth(pnx_ctx *ctx, bool /*identity_perm: h_perms holds the identity then*/) {
    const uint32_t G = ctx->n_groups, R = ctx->g_R, T = ctx->g_T, NB = ctx->n_blocks;
    int rc;
    const size_t out_n = (size_t)R * T * G;
    PNX_HIP(ctx, hipMemsetAsync(ctx->d_growth_out.p, 0, (out_n ? out_n : 1) * sizeof(uint64_t), ctx->stream));
    if (G == 0 || NB == 0) return PNX_OK;

    // host copies of the thresholds: cov_thr[T] then is_q0[T]
    const std::vector<uint32_t> &meta = ctx->h_thr_meta;
    const std::vector<uint32_t> &qt = ctx->h_qtab, &pm = ctx->h_perms;
    if (meta.size() != 2 * (size_t)T || qt.size() != (size_t)T * G || pm.size() != (size_t)R * G)
        return ctx->fail(PNX_EINVAL, "internal: growth tables missing");

    // distinct coverage thresholds >= 2 need a mask; c <= 1 is implied by presence
    std::vector<uint32_t> cvals;
    std::vector<int32_t> mask_of(T, -1);
    for (uint32_t t = 0; t < T; ++t) {
        if (meta[t] <= 1) continue;
        size_t k = 0;
        while (k < cvals.size() && cvals[k] != meta[t]) ++k;
        if (k == cvals.size()) cvals.push_back(meta[t]);
        mask_of[t] = (int32_t)k;
    }
    if (cvals.size() > 32) return ctx->fail(PNX_ELIMIT, "at most 32 distinct coverage thresholds >= 2 per call");

    // q > 0 pairs whose table rises by 0 or 1 per rank (always true for q in [0,1]) take the
    // slack form; anything else falls back to the plane-by-plane comparison kernel
    std::vector<uint32_t> q0, qslack, qgeneral, dtab((size_t)T * G, 0);
    for (uint32_t t = 0; t < T; ++t) {
        if (meta[T + t]) { q0.push_back(t); continue; }
        bool unit = true;
        uint32_t prev = 0;
        for (uint32_t j = 0; j < G && unit; ++j) {
            const uint32_t v = qt[(size_t)t * G + j];
            unit = v >= prev && v - prev <= 1;
            dtab[(size_t)t * G + j] = v - prev;
            prev = v;
        }
        (unit ? qslack : qgeneral).push_back(t);
    }
    const uint64_t row_words = (uint64_t)NB * BLOCK_WORDS;
    const uint64_t m_bytes = (uint64_t)G * row_words * 4;
    const bool use_fused = m_bytes < (1ull << 32);
    if (!use_fused) {
        // presence matrix >= 4 GiB: the 32-bit row offsets do not reach; every pair takes the
        // comparison kernel (64-bit addressing)
        for (uint32_t t : q0) qgeneral.push_back(t);
        for (uint32_t t : qslack) qgeneral.push_back(t);
        q0.clear();
        qslack.clear();
    }
    std::vector<uint32_t> is_delta(T, 0);
    for (uint32_t t : q0) is_delta[t] = 1;
    if (ctx->weighted)  // bp: the fused kernel keeps the slack-form pairs in delta form too
        for (uint32_t t : qslack) is_delta[t] = 1;

    // ONE upload of every small table of the call, staged in a vector that lives until the next
    // growth call: dmask[T][G] (0 / ~0 per rank) | rowoff[R][G] (byte offset of the row of rank j) |
    // is_delta[T] | cvals[32]
    std::vector<uint32_t> &tabs_h = ctx->h_growth_tabs;
    const size_t o_rowoff = (size_t)T * G, o_isd = o_rowoff + (size_t)R * G, o_cv = o_isd + T;
    tabs_h.assign(o_cv + 32, 0u);
    for (size_t i = 0; i < (size_t)T * G; ++i) tabs_h[i] = dtab[i] ? 0xFFFFFFFFu : 0u;
    if (use_fused)
        for (size_t i = 0; i < pm.size(); ++i) tabs_h[o_rowoff + i] = (uint32_t)((uint64_t)pm[i] * row_words * 4);
    for (uint32_t t = 0; t < T; ++t) tabs_h[o_isd + t] = is_delta[t];
    for (size_t k = 0; k < cvals.size(); ++k) tabs_h[o_cv + k] = cvals[k];
    if ((rc = ensure(ctx, ctx->d_thr_meta, tabs_h.size() * sizeof(uint32_t) + 16))) return rc;
    PNX_HIP(ctx, hipMemcpyAsync(ctx->d_thr_meta.p, tabs_h.data(), tabs_h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    const uint32_t *d_dmask = (const uint32_t *)ctx->d_thr_meta.p;
    const uint32_t *d_rowoff = d_dmask + o_rowoff;
    const uint32_t *d_isd = d_dmask + o_isd;
    const uint32_t *d_cvals = d_dmask + o_cv;
    if (!qgeneral.empty()) {  // the comparison kernel reads the order and the quorum tables themselves
        if ((rc = ensure(ctx, ctx->d_perms, pm.size() * sizeof(uint32_t)))) return rc;
        PNX_HIP(ctx, hipMemcpyAsync(ctx->d_perms.p, pm.data(), pm.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        PNX_HIP(ctx, hipMemcpyAsync(ctx->d_qtab.p, qt.data(), qt.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    }

    prof_begin(ctx, PNX_K_MASK);
    // layout of d_cmask: [n_c masks][NB][64] u32, then one scratch word (max weight)
    const size_t mask_words = cvals.size() * (size_t)NB * BLOCK_WORDS;
    if ((rc = ensure(ctx, ctx->d_cmask, (mask_words + 64) * sizeof(uint32_t)))) return rc;
    uint32_t *d_aux = (uint32_t *)ctx->d_cmask.p + mask_words;
    if (!cvals.empty())
        hipLaunchKernelGGL(k_cov_masks, dim3((NB + 3) / 4), dim3(256), 0, ctx->stream,
                           (const uint32_t *)ctx->d_countable_done->p, ctx->n_items, NB, d_cvals,
                           (uint32_t)cvals.size(), (uint32_t *)ctx->d_cmask.p);
    if (ctx->weighted && (rc = ensure_weight_planes(ctx, d_aux))) return rc;
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());

    // geometry: chunk-major grid (all R orders of a block chunk are neighbours in blockIdx)
    // enough workgroups for several rounds over the chip's wave slots (a short last round costs little
    // then), but not so many that the per-workgroup work -- clearing and flushing the LDS accumulators --
    // shows: R orders x n_chunks ~ 8 x (CUs x 6 workgroups of 4 waves)
    uint32_t target_wgs = 8u * (uint32_t)ctx->prop.multiProcessorCount * 6u;
    if (const char *e = std::getenv("PNX_GROWTH_WGS")) target_wgs = (uint32_t)std::atoi(e);  // experiments
    uint32_t n_chunks = std::max<uint32_t>(target_wgs / R, (NB + 63) / 64);
    if (n_chunks < 1) n_chunks = 1;
    const uint32_t max_chunks = (NB + GROW_WAVES - 1) / GROW_WAVES;
    if (n_chunks > max_chunks) n_chunks = max_chunks;
    // blocks per chunk: a multiple of the waves of a workgroup (a wave takes every GROW_WAVES-th block of its chunk: 7 blocks
    // for 4 waves leave one wave idle for an eighth of the kernel -- 2.29 -> 2.0x ms for 16 orders of cfg4)
    uint32_t bpc = (NB + n_chunks - 1) / n_chunks;
    if (!std::getenv("PNX_GROWTH_BPC_ANY")) bpc = (bpc + GROW_WAVES - 1) / GROW_WAVES * GROW_WAVES;
    n_chunks = (NB + bpc - 1) / bpc;
    const size_t wp_bytes = ctx->weighted ? (size_t)GROW_WAVES * WPLANES_MAX * 64 * sizeof(uint32_t) : 0;  // comparison kernel
    const bool w16 = ctx->weighted && ctx->n_wplanes <= 16;  // every weight < 2^16
    const size_t wl_bytes = ctx->weighted && GROW_W_LDS ? (size_t)GROW_WAVES * 2048 * (w16 ? 2 : 4) : 0;   // fused kernel: staged weights
    const uint32_t *d_wpl = ctx->weighted ? (const uint32_t *)ctx->d_wplanes.p : nullptr;
    const uint32_t n_planes = ctx->weighted ? ctx->n_wplanes : 0;

    uint32_t bits = 1;
    while (bits < 32 && (G >> bits) != 0) ++bits;
    // PNX_GROWTH_STEP=1: two ranks per step by the general rule even where every pair of a table is d = (1, 0)  (tests)
    const char *e_step = std::getenv("PNX_GROWTH_STEP");
    const bool general_step = e_step && e_step[0] == '1';
    // does the table of pair t rise at every even rank and only there, over the full batches (q = 0.5)?
    auto alternates = [&](uint32_t t) {
        const uint32_t full = G / GROW_PREFETCH * GROW_PREFETCH;
        for (uint32_t j = 0; j < full; ++j)
            if ((dtab[(size_t)t * G + j] != 0) != ((j & 1u) == 0)) return false;
        return true;
    };
    prof_begin(ctx, PNX_K_GROWTH);
    {   // fused launches: up to GROW_Q0_MAX q == 0 pairs + up to 2 slack pairs each
        size_t i0 = 0, iq = 0;
        while (i0 < q0.size() || iq < qslack.size()) {
            GrowthTabs tabs;
            const int n0 = (int)std::min<size_t>(GROW_Q0_MAX, q0.size() - i0);
            for (int k = 0; k < GROW_Q0_MAX; ++k) {
                tabs.q0_midx[k] = k < n0 ? mask_of[q0[i0 + k]] : -1;
                tabs.q0_slot[k] = k < n0 ? q0[i0 + k] : 0;
            }
            const int nq = (int)std::min<size_t>(2, qslack.size() - iq);
            for (int k = 0; k < 2; ++k) {
                tabs.qq_midx[k] = k < nq ? mask_of[qslack[iq + k]] : -1;
                tabs.qq_slot[k] = k < nq ? qslack[iq + k] : 0;
            }
            bool alt = nq > 0 && !general_step;
            for (int k = 0; k < nq; ++k)
                if (!alternates(qslack[iq + k])) alt = false;
            // bp: the flip events a wave queues before its 64 lanes apply them, one event per lane: a longer queue is applied in
            // full rounds (128: a flush of 65 .. 128 events is one full round and a partial one)
            uint32_t evq_n = GROW_EVQ;
            if (const char *e = std::getenv("PNX_GROWTH_EVQ")) evq_n = std::max(128, std::atoi(e)) / 64 * 64;  // experiments
            const size_t evq_bytes = ctx->weighted ? (size_t)GROW_WAVES * evq_n * (1 + n0 + 2 * nq) * 4 : 0;
            const size_t shmem = (size_t)(n0 + nq) * G * 8 + (size_t)GROW_WAVES * (n0 + nq) * (GROW_PREFETCH / 2) * 4 + wl_bytes + evq_bytes;
            if (shmem > 150 * 1024)
                return ctx->fail(PNX_ELIMIT, "ordered growth: %u groups x %d threshold pairs exceed the LDS accumulators",
                                 G, n0 + nq);
            auto go = [&](auto kern) {
                if (shmem > 64 * 1024)
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                hipLaunchKernelGGL(kern, dim3(R * n_chunks), dim3(GROW_WAVES * 64), shmem, ctx->stream,
                                   (const uint32_t *)ctx->d_M.p, NB, G, d_rowoff, R, bpc, (const uint32_t *)ctx->d_cmask.p,
                                   tabs, d_dmask, T, (const uint32_t *)ctx->d_weights.p, ctx->n_items,
                                   (unsigned long long *)ctx->d_growth_out.p, evq_n);
            };
#define PNX_GROW_NQ(NPL1, N0V, W)                                                                 \
    do {                                                                                          \
        if (nq == 0) go(k_growth_fused<NPL1, N0V, 0, W, false>);                                  \
        else if (nq == 1 && alt) go(k_growth_fused<NPL1, N0V, 1, W, true>);                       \
        else if (nq == 1) go(k_growth_fused<NPL1, N0V, 1, W, false>);                             \
        else if (alt) go(k_growth_fused<NPL1, N0V, 2, W, true>);                                  \
        else go(k_growth_fused<NPL1, N0V, 2, W, false>);                                          \
    } while (0)
#define PNX_GROW_N0(NPL1, W)                                                                      \
    do {                                                                                          \
        switch (n0) {                                                                             \
            case 0: PNX_GROW_NQ(NPL1, 0, W); break;                                               \
            case 1: PNX_GROW_NQ(NPL1, 1, W); break;                                               \
            case 2: PNX_GROW_NQ(NPL1, 2, W); break;                                               \
            case 3: PNX_GROW_NQ(NPL1, 3, W); break;                                               \
            default: PNX_GROW_NQ(NPL1, 4, W); break;                                              \
        }                                                                                         \
    } while (0)
#define PNX_GROW_DISPATCH(NPL1)                                                                   \
    do {                                                                                          \
        if (!ctx->weighted) PNX_GROW_N0(NPL1, 0);                                                 \
        else if (w16) PNX_GROW_N0(NPL1, 1);                                                       \
        else PNX_GROW_N0(NPL1, 2);                                                                \
    } while (0)
            // planes for s in [-G, G]: bits(G) + 1 (sign)
            if (bits + 1 <= 9) PNX_GROW_DISPATCH(9);
            else if (bits + 1 <= 11) PNX_GROW_DISPATCH(11);
            else if (bits + 1 <= 14) PNX_GROW_DISPATCH(14);
            else if (bits + 1 <= 17) PNX_GROW_DISPATCH(17);
            else return ctx->fail(PNX_ELIMIT, "ordered growth supports at most 65535 groups");
#undef PNX_GROW_DISPATCH
#undef PNX_GROW_N0
#undef PNX_GROW_NQ
            i0 += (size_t)n0;
            iq += (size_t)nq;
        }
    }
    for (uint32_t t : qgeneral) {  // arbitrary quorum tables: comparison kernel, one launch each
        if ((size_t)G * 8 + wp_bytes > 64 * 1024)
            return ctx->fail(PNX_ELIMIT, "ordered growth supports at most %u groups (got %u)",
                             (unsigned)((64 * 1024 - wp_bytes) / 8), G);
        const uint32_t *d_mask_t = mask_of[t] >= 0 ? (const uint32_t *)ctx->d_cmask.p + (size_t)mask_of[t] * NB * BLOCK_WORDS : nullptr;
        const uint32_t *d_q = (const uint32_t *)ctx->d_qtab.p + (size_t)t * G;
        const size_t shmem = (size_t)G * 8 + wp_bytes;
        // this kernel numbers workgroups order-major: r = blockIdx / n_chunks
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(R * n_chunks), dim3(GROW_WAVES * 64), shmem, ctx->stream,
                               (const uint32_t *)ctx->d_M.p, row_words, NB, G, (const uint32_t *)ctx->d_perms.p,
                               n_chunks, bpc, d_mask_t, d_q, t, T, d_wpl, n_planes,
                               (unsigned long long *)ctx->d_growth_out.p);
        };
        if (ctx->weighted) {
            if (bits <= 8) go(k_growth_quorum<8, true>);
            else if (bits <= 12) go(k_growth_quorum<12, true>);
            else go(k_growth_quorum<16, true>);
        } else {
            if (bits <= 8) go(k_growth_quorum<8, false>);
            else if (bits <= 12) go(k_growth_quorum<12, false>);
            else go(k_growth_quorum<16, false>);
        }
    }
    // deltas -> running sums (one wave per (order, pair) row)
    hipLaunchKernelGGL(k_growth_prefix, dim3(R * T), dim3(64), 0, ctx->stream,
                       (unsigned long long *)ctx->d_growth_out.p, G, T, d_isd);
    prof_end(ctx);
    PNX_HIP(ctx, hipGetLastError());
    return PNX_OK;
}

}  // namespace pnx

namespace pnx {
// pnx_preload (see kernels_gfa.hip): touching one kernel loads the code object of this translation unit
void preload_growth(unsigned what) {
    hipFuncAttributes a;
    auto touch = [&a](const void *k) { (void)hipFuncGetAttributes(&a, k); };
    if (what & PNX_PRELOAD_GROWTH) {
        touch((const void *)k_cov_masks);
        touch((const void *)k_growth_prefix);
        touch((const void *)k_max_u32);
    }
}
}  // namespace pnx
