// upload_scan.hip -- what the ONE read of the steps at upload leaves behind, for the one-shot route (kernels_band.hip): the ids
// validated, the chunks of every path summarised, the paths cut where they turn round or jump back, and the paths that do not
// follow the ids at all stored a second time in the order of the ids.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>  // rocprim's texture iterator calls memset
#include <vector>

#include <rocprim/rocprim.hpp>

#include "pnx_context.hpp"
#include "step_chunks.hpp"

namespace pnx {

// ------------------------------------------------------------------------------------------
// upload validation of a graph the one-shot route may take: the same one read of the steps also says where its paths
// turn round or jump back
// ------------------------------------------------------------------------------------------
// The paths of a pangenome run through the ids of a sorted graph in order -- except at LARGE rearrangements: an inversion
// (the path runs downwards for a stretch), a duplication or a translocation (the ids jump back and run on).  No single
// position per band edge deals the steps of such a path to bands (kernels_band.hip: a whole inverted stretch would fall to the
// bands at its two ends and be spilled), but every PIECE between two such breaks is a path that follows the ids.  So the
// read of the steps that validates the ids at upload (the reference panics on unknown nodes, graph_broker/util.rs:1021) also
// leaves a summary of every chunk of 4096 steps -- five evenly spaced ids, and how many steps go up / down --, and the host
// cuts every path where the summaries turn round or jump back by more than two bands for at least two chunks
// (path_cuts_from_chunks; refine_path_cuts then moves every cut to the step where the ids jump).  The one-shot pass then takes the pieces as entries of their own in the visiting order, under
// their path's group: AbacusByTotal::coverage (abacus.rs:727-742) counts a group once per item however its steps are split.
// Small disorder (pansyn-v1r's 64-step blocks) never makes a cut: it is spilled as before.
__global__ __launch_bounds__(256) void k_chunk_summaries(const uint32_t *__restrict__ items, const uint64_t *__restrict__ path_off,
                                                         const uint64_t *__restrict__ chunk_off, uint32_t n_paths, uint64_t n_chunks,
                                                         uint32_t n_items, ChunkSummary *__restrict__ out, uint32_t *__restrict__ bad) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const RunChunk ch = chunk_of(c, chunk_off, path_off, n_paths);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t a0 = ch.start & ~3ull;
    const uint32_t head = (uint32_t)(ch.start - a0), nal = head + ch.len;
    uint32_t up = 0, down = 0;
    bool b = false;
    for (uint32_t r0 = 0; r0 < nal; r0 += 1024u) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t q = r0 + (uint32_t)u * 256u + lane * 4u;
            v[u] = q < nal ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(items + a0 + q)) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t q = r0 + (uint32_t)u * 256u + lane * 4u;
            const uint32_t ids[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = q + (uint32_t)e - head < ch.len;  // (unsigned: false before the chunk too)
                if (in && ids[e] - 1u >= n_items) b = true;
                if (e < 3 && in && q + (uint32_t)e + 1u - head < ch.len) {  // the steps inside one 16-byte load: three of every four
                    up += ids[e + 1] > ids[e] ? 1u : 0u;
                    down += ids[e + 1] < ids[e] ? 1u : 0u;
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        up += __shfl_down(up, o);
        down += __shfl_down(down, o);
    }
    if (__ballot(b) && lane == 0) atomicOr(bad, 1u);
    if (lane < 5) out[c].s[lane] = items[ch.start + (uint64_t)(ch.len - 1u) * lane / 4u];
    if (lane == 0) {
        out[c].up = up;
        out[c].down = down;
    }
}

// A cut found from the summaries lies at a chunk boundary; the break itself -- the step where the ids jump -- is somewhere in
// the chunk before it or the one behind.  One wave per cut looks at those 8192 steps and moves the cut to the largest jump
// between two consecutive steps (a path that follows the ids moves by a few ids per step; at a break by more than two bands).
__global__ __launch_bounds__(64) void k_refine_cuts(const uint32_t *__restrict__ items, const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi,
                                                    uint64_t *__restrict__ cuts) {
    const uint32_t lane = threadIdx.x;
    const uint64_t a = lo[blockIdx.x], z = hi[blockIdx.x];  // pairs (j - 1, j) for j in [a + 1, z)
    uint32_t best = 0;
    uint64_t at = cuts[blockIdx.x];
    for (uint64_t j = a + 1 + lane; j < z; j += 64) {
        const uint32_t x = items[j - 1], y = items[j];
        const uint32_t d = x > y ? x - y : y - x;
        if (d > best) {
            best = d;
            at = j;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t ob = __shfl_xor(best, o);
        const uint64_t oa = __shfl_xor(at, o);
        if (ob > best || (ob == best && oa < at)) {
            best = ob;
            at = oa;
        }
    }
    if (lane == 0) cuts[blockIdx.x] = at;
}

int refine_path_cuts(pnx_ctx *ctx) {
    const size_t n = ctx->h_cuts.size();
    if (!n) return PNX_OK;
    std::vector<uint64_t> lo(n), hi(n);
    for (uint32_t p = 0; p < ctx->n_paths; ++p)
        for (uint32_t c = ctx->h_cut_off[p]; c < ctx->h_cut_off[p + 1]; ++c) {
            const uint64_t cut = ctx->h_cuts[c], ps = ctx->h_path_off[p], pe = ctx->h_path_off[p + 1];
            lo[c] = cut - ps > RUN_CHUNK ? cut - RUN_CHUNK : ps;
            hi[c] = pe - cut > RUN_CHUNK ? cut + RUN_CHUNK : pe;
        }
    int rc;
    DevBuf scratch;
    if ((rc = ensure(ctx, scratch, 3 * n * 8))) return rc;
    uint64_t *d = (uint64_t *)scratch.p;
    hipError_t e = hipMemcpyAsync(d, lo.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + n, hi.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + 2 * n, ctx->h_cuts.data(), n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_refine_cuts, dim3((unsigned)n), dim3(64), 0, ctx->stream, (const uint32_t *)ctx->d_items.p, (const uint64_t *)d,
                           (const uint64_t *)(d + n), d + 2 * n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_cuts.data(), d + 2 * n, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    release(scratch);
    if (e != hipSuccess) return ctx->fail(PNX_EHIP, "moving the cuts of the paths to their steps failed: %s", hipGetErrorString(e));
    return PNX_OK;
}

int launch_chunk_summaries(pnx_ctx *ctx, uint32_t *d_bad) {
    if (ctx->n_steps == 0) return PNX_OK;
    int rc;
    if ((rc = ensure_chunk_off(ctx))) return rc;
    const uint64_t n_chunks = ctx->h_chunk_off[ctx->n_paths];
    if ((n_chunks + 3) / 4 > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "too many path chunks");
    if ((rc = ensure(ctx, ctx->d_chunk_sum, n_chunks * sizeof(ChunkSummary)))) return rc;
    hipLaunchKernelGGL(k_chunk_summaries, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0, ctx->stream, (const uint32_t *)ctx->d_items.p,
                       (const uint64_t *)ctx->d_path_off.p, (const uint64_t *)ctx->d_chunk_off.p, ctx->n_paths, n_chunks, ctx->n_items,
                       (ChunkSummary *)ctx->d_chunk_sum.p, d_bad);
    PNX_HIP(ctx, hipGetLastError());
    ctx->h_chunk_sum.resize(n_chunks);
    PNX_HIP(ctx, hipMemcpyAsync(ctx->h_chunk_sum.data(), ctx->d_chunk_sum.p, n_chunks * sizeof(ChunkSummary), hipMemcpyDeviceToHost, ctx->stream));
    return PNX_OK;
}

// Where the paths are cut (absolute step positions, chunk boundaries): h_cut_off[p] .. h_cut_off[p + 1] into h_cuts.
// A chunk lies between the second smallest and the second largest of its five ids (one id from elsewhere does not move
// it) and runs the way most of its steps go; a chunk whose steps go both ways (a quarter against the rest) says nothing.
// A cut in front of chunk c: c and the chunk behind it run on from each other, and c does not run on from the
// chunk before it -- it runs the other way, or it begins more than two bands behind where that chunk began.
void path_cuts_from_chunks(pnx_ctx *ctx) {
    constexpr uint32_t SLACK = 2u << 13, MIN_RUN = 2, MAX_CUTS = 62;
    const uint32_t P = ctx->n_paths;
    ctx->h_cut_off.assign((size_t)P + 1, 0);
    ctx->h_cuts.clear();
    ctx->h_jumbled.assign(P, 0);
    if (ctx->h_chunk_sum.empty() || ctx->h_chunk_off.size() != (size_t)P + 1) return;
    struct Ck {
        uint32_t lo, hi;
        int dir;
    };
    std::vector<Ck> ck;
    for (uint32_t p = 0; p < P; ++p) {
        const uint64_t c0 = ctx->h_chunk_off[p], c1 = ctx->h_chunk_off[p + 1];
        const size_t first_cut = ctx->h_cuts.size();
        {   // a path most of whose chunks go both ways follows the ids nowhere (sort_jumbled_paths)
            uint64_t both = 0;
            for (uint64_t c = c0; c < c1; ++c) {
                const ChunkSummary &s = ctx->h_chunk_sum[c];
                const uint32_t mn = s.up < s.down ? s.up : s.down, all = s.up + s.down;
                both += (all > 64 && (uint64_t)mn * 4 > all) ? 1 : 0;
            }
            ctx->h_jumbled[p] = c1 > c0 && both * 2 > c1 - c0 && ctx->h_path_off[p + 1] - ctx->h_path_off[p] >= 1024 ? 1 : 0;
        }
        if (c1 - c0 >= 2 * MIN_RUN && !ctx->h_jumbled[p]) {
            ck.resize(c1 - c0);
            for (uint64_t c = c0; c < c1; ++c) {
                const ChunkSummary &s = ctx->h_chunk_sum[c];
                uint32_t v[5] = {s.s[0], s.s[1], s.s[2], s.s[3], s.s[4]};
                std::sort(v, v + 5);
                const uint32_t mn = s.up < s.down ? s.up : s.down, all = s.up + s.down;
                const int dir = (all > 64 && (uint64_t)mn * 4 > all) ? 0 : (s.up > s.down ? 1 : (s.down > s.up ? -1 : 0));
                ck[c - c0] = Ck{v[1], v[3], dir};
            }
            auto runs_on = [&](const Ck &a, const Ck &b) {  // b continues a
                if (a.dir == 0 || b.dir == 0) return true;
                if (a.dir != b.dir) return false;
                return a.dir > 0 ? (uint64_t)b.lo + SLACK >= a.lo : (uint64_t)a.hi + SLACK >= b.hi;
            };
            const size_t n = ck.size();
            size_t last = 0;  // the chunk the current piece was last seen to run on from
            for (size_t i = 1; i + MIN_RUN <= n; ++i) {
                if (ck[i].dir == 0) continue;
                if (runs_on(ck[last], ck[i])) {
                    last = i;
                    continue;
                }
                bool fresh = true;  // a new piece: it holds together for MIN_RUN chunks
                for (size_t j = i + 1; j < i + MIN_RUN && fresh; ++j) fresh = ck[j].dir != 0 && runs_on(ck[j - 1], ck[j]) && !runs_on(ck[last], ck[j]);
                if (fresh) {
                    ctx->h_cuts.push_back(ctx->h_path_off[p] + (uint64_t)i * RUN_CHUNK);
                    last = i;
                }
            }
            if (ctx->h_cuts.size() - first_cut > MAX_CUTS) ctx->h_cuts.resize(first_cut);  // (a path in pieces all over: no pieces)
        }
        ctx->h_cut_off[p + 1] = (uint32_t)ctx->h_cuts.size();
    }
}

// A path that follows the ids NOWHERE (shuffled; most of its chunks go both ways) has no pieces to cut it into.  But
// AbacusByTotal::coverage (abacus.rs:727-742) counts a group once per item whatever the order of its steps: the same steps in
// the order of the ids are the same path to every result of the hot path.  So such a path is stored a SECOND time, sorted
// (a radix sort of its ids, once, at upload), behind the steps of the graph -- the ItemTable the caller uploaded stays what
// it was (pnx_get_csr, the path rows) --, and the one-shot pass takes the copy as the entry of that path: one read at the rate
// of any other path, where marking its steps one atomic at a time costs 0.12 ms per 2.5 M steps (kernels_band.hip: BandLoose,
// which stays for the paths that are only partly jumbled).  Bounded: all such paths together may hold an eighth of the
// steps (a graph of shuffled paths takes the path rows); the copy costs 4 bytes per step of those paths.
// -> ctx->h_sorted_at[p] = index of the copy's first step in d_items (>= n_steps), or 0.
int sort_jumbled_paths(pnx_ctx *ctx) {
    const uint32_t P = ctx->n_paths;
    ctx->h_sorted_at.assign(P, 0);
    if (ctx->h_jumbled.size() != P) return PNX_OK;
    const uint64_t S = ctx->n_steps;
    uint64_t total = 0, extra = 0;
    uint32_t n = 0;
    for (uint32_t p = 0; p < P; ++p)
        if (ctx->h_jumbled[p]) {
            const uint64_t len = ctx->h_path_off[p + 1] - ctx->h_path_off[p];
            total += len;
            extra += (len + 63) & ~63ull;  // (every copy begins at a multiple of 64 steps: sectors and 16-byte loads stay aligned)
            ++n;
        }
    if (!n || total > std::max<uint64_t>(S / 8, 4ull << 20)) return PNX_OK;
    const uint64_t base = (S + 63) & ~63ull;
    if (base + extra >= (1ull << 40)) return PNX_OK;
    // the steps move into a buffer with room for the copies behind them
    DevBuf bigger;
    int rc;
    if ((rc = ensure(ctx, bigger, (base + extra) * 4 + 64))) return rc;
    // (no early-return macro from here on: `bigger` -- a copy of all the steps -- and `tmp` are released on every way out)
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == PNX_OK) rc = ctx->fail(e == hipErrorOutOfMemory ? PNX_ENOMEM : PNX_EHIP, "%s failed: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    hip_ok(hipMemcpyAsync(bigger.p, ctx->d_items.p, S * 4, hipMemcpyDeviceToDevice, ctx->stream), "copying the steps at upload");
    if (rc == PNX_OK && base > S) hip_ok(hipMemsetAsync((uint32_t *)bigger.p + S, 0, (base - S) * 4, ctx->stream), "padding the steps at upload");
    unsigned bits = 1;
    while (bits < 32 && (ctx->n_items >> bits)) ++bits;
    DevBuf tmp;
    uint64_t at = base;
    for (uint32_t p = 0; p < P && rc == PNX_OK; ++p) {
        if (!ctx->h_jumbled[p]) continue;
        const uint64_t ps = ctx->h_path_off[p], len = ctx->h_path_off[p + 1] - ps;
        const uint32_t *in = (const uint32_t *)bigger.p + ps;
        uint32_t *out = (uint32_t *)bigger.p + at;
        size_t bytes = 0;
        hipError_t e = rocprim::radix_sort_keys(nullptr, bytes, in, out, (size_t)len, 0u, bits, ctx->stream);
        if (e == hipSuccess && (rc = ensure(ctx, tmp, bytes ? bytes : 8)) == PNX_OK) e = rocprim::radix_sort_keys(tmp.p, bytes, in, out, (size_t)len, 0u, bits, ctx->stream);
        if (e != hipSuccess) rc = ctx->fail(PNX_EHIP, "sorting a path at upload failed: %s", hipGetErrorString(e));
        const uint64_t pad = ((len + 63) & ~63ull) - len;
        if (rc == PNX_OK && pad) hip_ok(hipMemsetAsync(out + len, 0, pad * 4, ctx->stream), "padding a sorted path");
        ctx->h_sorted_at[p] = at;
        at += len + pad;
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == PNX_OK) rc = ctx->fail(PNX_EHIP, "sorting a path at upload failed");
    release(tmp);
    if (rc != PNX_OK) {
        release(bigger);
        ctx->h_sorted_at.assign(P, 0);
        return rc;
    }
    release(ctx->d_items);
    ctx->d_items = bigger;
    ctx->n_sorted_copies = n;
    return PNX_OK;
}

}  // namespace pnx
